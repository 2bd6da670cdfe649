"""emote_hack_amd - MI355X-native diffusion hot path of johndpope/Emote-hack.

Host-side mirror of the reference's Python surface for that path (UNet3DConditionModel,
ReferenceAttentionControl, EMOAnimationPipeline, schedulers, context windows) over hand-written
gfx950 HIP kernels reached through the C ABI in include/emo_hip.h.  No CPU fallback.
"""
from .config import normalize_unet_config, unet_config_from_yaml  # noqa: F401
from .context import get_context_scheduler, ordered_halving, uniform  # noqa: F401
from .scheduler import DDIMScheduler, DDPMScheduler  # noqa: F401


def __getattr__(name):  # heavy modules on demand
    if name in ("UNet3DConditionModel", "UNet3DConditionOutput"):
        from . import unet
        return getattr(unet, name)
    if name == "ReferenceAttentionControl":
        from .reference_control import ReferenceAttentionControl
        return ReferenceAttentionControl
    if name in ("EMOAnimationPipeline", "AnimationPipelineOutput"):
        from . import pipeline
        return getattr(pipeline, name)
    if name in ("ControlNetModel", "ControlNetOutput"):
        from . import controlnet
        return getattr(controlnet, name)
    if name == "AppearanceEncoderModel":
        from .appearance_encoder import AppearanceEncoderModel
        return AppearanceEncoderModel
    raise AttributeError(name)
