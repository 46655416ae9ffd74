"""devias_amd: MI355X-native (gfx950 HIP) implementation of the DEVIAS slot-ViT training step.

Importing the package is cheap and GPU-free; the HIP library is loaded on first kernel use and its absence is an
error (there is no CPU or eager-PyTorch fallback for the hot path)."""
__version__ = "0.1.0"

__all__ = ["create_model", "VisionTransformer", "TrainLoss", "train_class_batch"]


def __getattr__(name):
    if name in ("create_model", "VisionTransformer", "slot_vit_base_patch16_224", "slot_vit_small_patch16_224",
                "slot_vit_large_patch16_224"):
        from . import modeling_slot
        return getattr(modeling_slot, name)
    if name == "TrainLoss":
        from .train_loss import TrainLoss
        return TrainLoss
    if name == "train_class_batch":
        from .engine_for_slot import train_class_batch
        return train_class_batch
    raise AttributeError(name)
