// ROCTX ranges around the fused regions, the loss and the optimizer (visible with `rocprofv3 --marker-trace`): the reference has
// no profiler ranges (utils/utils.py:120-164 only keeps wall-clock meters).  libdevias_amd keeps its single link dependency on the HIP runtime:
// the marker library (rocprofiler-sdk's roctx, or the legacy libroctx64) is resolved with dlopen at the first range, and only when the
// environment variable DEVIAS_ROCTX is set to a non-zero value -- otherwise a range is one predictable branch.
#pragma once
void devias_roctx_push(const char* name);
void devias_roctx_pop(void);
bool devias_roctx_enabled(void);
struct devias_range {
    bool on;
    explicit devias_range(const char* name) : on(devias_roctx_enabled()) { if (on) devias_roctx_push(name); }
    ~devias_range() { if (on) devias_roctx_pop(); }
};
