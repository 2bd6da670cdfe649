"""AppearanceEncoderModel - the ReferenceNet used by the pipeline
(magicanimate/models/appearance_encoder.py:217-633,777-1066).

In the reference this is a diffusers 2-D UNet clone without conv_norm_out / conv_out whose output is
discarded: only the LN1 activations written to the banks matter, and the last transformer block of the up path is
reduced to what feeds its bank (norm, proj_in, norm1; :613-621) - same key set here (tests/golden/ints.json
"appearance_encoder_gutted", extracted from the reference ctor by AST), and the pass stops at that bank write.  Its block arithmetic lives in
third-party diffusers (parity unpinned).  Here it is the F=1, no-motion-module instance
of the SAME kernels as the Backbone (ResnetBlock3D at F=1 == 2-D resnet, Transformer3D at F=1 ==
Transformer2D), with identical state-dict key names.  forward() accepts the 2-D call of the pipeline
(`appearance_encoder(ref_latents (N,4,h,w), t, encoder_hidden_states=...)`, EMOAnimationPipeline.py:711-716).
"""
from __future__ import annotations

from .unet import UNet3DConditionModel


class AppearanceEncoderModel(UNet3DConditionModel):
    def __init__(self, **kwargs):
        kwargs = dict(kwargs)
        for k in ("use_motion_module", "motion_module_type", "motion_module_kwargs"):
            kwargs.pop(k, None)
        kwargs.setdefault("unet_use_cross_frame_attention", False)
        kwargs.setdefault("unet_use_temporal_attention", False)
        kwargs.setdefault("_has_out", False)
        # appearance_encoder.py:613-621: up_blocks[3].attentions[2] keeps norm / proj_in / norm1 only; a reference-shaped
        # ReferenceNet checkpoint has no other keys under that block, and the pass ends at its bank write
        kwargs.setdefault("_gut_last_transformer", True)
        super().__init__(**kwargs)

    def forward(self, sample, timestep, encoder_hidden_states, **kw):
        if sample.dim() == 4:
            sample = sample.unsqueeze(2)
        return super().forward(sample, timestep, encoder_hidden_states, **kw)

    __call__ = forward
