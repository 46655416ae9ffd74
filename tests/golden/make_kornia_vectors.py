#!/usr/bin/env python3
"""Known-answer vectors for the two kornia functions behind FAME (utils/transform/fame.py:20-22 GaussianBlur2d, :47 rgb_to_hsv of the
reference), computed here in plain Python float64 from kornia's PUBLISHED formulas only -- no torch, no oracle code, no kornia (absent
from this image and not version-pinned by the reference, docs/INSTALL.md:32):

  kornia.filters.get_gaussian_kernel1d(k, s):  g[i] = exp(-(i - k//2)^2 / (2 s^2)),  normalised to sum 1
  kornia.filters.gaussian_blur2d(x, (k,k), (s,s), border_type='reflect'):  separable correlation of the 'reflect'-padded image
      (torch 'reflect': index -j -> j, H-1+j -> H-1-j: the edge sample is not repeated)
  kornia.color.rgb_to_hsv(rgb in [0,1], eps=1e-8):  v = max;  s = (max-min)/(max+eps);
      h = 2*pi * (((g-b)/d) mod 6)/6 if max is r;  2*pi*((b-r)/d + 2)/6 if g;  2*pi*((r-g)/d + 4)/6 if b;   d = max-min (1 where 0)
  FAME colour bins (fame.py:57-64, which multiplies the RADIAN hue by 2*pi once more):  hx = (s cos(2 pi h) + 1)/2, hy = (s sin(2 pi h) + 1)/2,
      bin = round(9 hx + 1) + 10 (round(9 hy + 1) - 1) + 100 (round(9 v + 1) - 1)

Writes tests/golden/kornia_vectors.json.  Run: python tests/golden/make_kornia_vectors.py"""
import json
import math
import os

K, SIGMA = 11, 11.0 / 3.0          # FAME(crop_size=112): gauss_size = int(0.1*112)//2*2+1 = 11, sigma = 11/3  (fame.py:18-22)


def taps(k, s):
    g = [math.exp(-((i - k // 2) ** 2) / (2.0 * s * s)) for i in range(k)]
    t = sum(g)
    return [v / t for v in g]


def reflect(i, n):
    if i < 0:
        return -i
    if i >= n:
        return 2 * (n - 1) - i
    return i


def blur(img, k, s):
    H, W = len(img), len(img[0])
    g = taps(k, s)
    r = k // 2
    tmp = [[sum(g[j] * img[y][reflect(x + j - r, W)] for j in range(k)) for x in range(W)] for y in range(H)]       # along x
    return [[sum(g[j] * tmp[reflect(y + j - r, H)][x] for j in range(k)) for x in range(W)] for y in range(H)]     # along y


def rgb_to_hsv(r, g, b, eps=1e-8):
    mx, mn = max(r, g, b), min(r, g, b)
    d = mx - mn
    v = mx
    s = d / (mx + eps)
    dd = d if d != 0 else 1.0
    if mx == r:                      # first maximal channel, as torch.max does
        h = ((g - b) / dd) / 6.0
    elif mx == g:
        h = ((b - r) / dd + 2.0) / 6.0
    else:
        h = ((r - g) / dd + 4.0) / 6.0
    h = (h % 1.0) * 2.0 * math.pi
    return h, s, v


def fame_bin(r, g, b):
    h, s, v = rgb_to_hsv(r, g, b)
    hx = (s * math.cos(h * 2 * math.pi) + 1) / 2
    hy = (s * math.sin(h * 2 * math.pi) + 1) / 2
    q = [hx * 9 + 1, hy * 9 + 1, v * 9 + 1]
    margin = min(abs((t % 1.0) - 0.5) for t in q)          # distance of the nearest argument from a rounding boundary
    # torch.round is round-half-even; no listed colour sits on a boundary (margin reported)
    hb, sb, vb = (int(math.floor(t + 0.5)) for t in q)
    return hb + (sb - 1) * 10 + (vb - 1) * 100, margin


def main():
    out = {"ksize": K, "sigma": SIGMA, "taps": taps(K, SIGMA)}
    H = W = 16
    cases = []
    for (iy, ix) in ((8, 8), (1, 2), (0, 15), (15, 0)):
        img = [[0.0] * W for _ in range(H)]
        img[iy][ix] = 1.0
        cases.append({"impulse": [iy, ix], "shape": [H, W], "blurred": blur(img, K, SIGMA)})
    ramp = [[(3 * y + 5 * x) % 17 / 16.0 for x in range(W)] for y in range(H)]
    cases.append({"image": ramp, "shape": [H, W], "blurred": blur(ramp, K, SIGMA)})
    out["blur_cases"] = cases
    colours = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (0, 1, 1), (1, 0, 1), (1, 1, 1),          # the 8 cube corners
               (0.25, 0.25, 0.25), (0.5, 0.5, 0.5), (0.75, 0.75, 0.75),                                       # 3 greys
               (0.2, 0.6, 0.4), (0.9, 0.3, 0.1), (0.1, 0.2, 0.8), (0.55, 0.5, 0.05), (0.3, 0.85, 0.9)]        # generic
    hsv = []
    for c in colours:
        h, s, v = rgb_to_hsv(*[float(t) for t in c])
        b, m = fame_bin(*[float(t) for t in c])
        hsv.append({"rgb": list(map(float, c)), "h": h, "s": s, "v": v, "fame_bin": b, "bin_margin": m})
    out["hsv_cases"] = hsv
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kornia_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes; min bin margin", min(c["bin_margin"] for c in hsv))


if __name__ == "__main__":
    main()
