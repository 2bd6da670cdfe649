"""CPU restatement of the ControlNet branch (magicanimate/models/controlnet.py) - TEST INFRASTRUCTURE ONLY
(imported by tests/, never by the product path).

In-tree arithmetic restated here: ControlNetConditioningEmbedding.forward (:78-91), the zero-convolutions and the
residual scaling of ControlNetModel.forward (:523-559).  The down / mid blocks are diffusers 2-D blocks (not in the
tree: parity unpinned, like the AppearanceEncoder); they are realised as the F=1 instance of the pinned 3-D blocks of
oracle/unet_ref.py.  The conditioning embedding is pinned by tests/golden/controlnet.safetensors (the reference class
run through tools/oracle/gen_golden.py)."""
import torch
import torch.nn.functional as F

from . import unet_ref as U


def cond_embedding(sd, cond, prefix="controlnet_cond_embedding"):
    """controlnet.py:78-91: conv_in -> silu -> [conv -> silu]* -> conv_out."""
    x = F.silu(F.conv2d(cond, sd[f"{prefix}.conv_in.weight"], sd[f"{prefix}.conv_in.bias"], padding=1))
    i = 0
    while f"{prefix}.blocks.{i}.weight" in sd:
        x = F.silu(F.conv2d(x, sd[f"{prefix}.blocks.{i}.weight"], sd[f"{prefix}.blocks.{i}.bias"], padding=1, stride=2 if i % 2 else 1))
        i += 1
    return F.conv2d(x, sd[f"{prefix}.conv_out.weight"], sd[f"{prefix}.conv_out.bias"], padding=1)


def controlnet_forward(sd, cfg, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0):
    """controlnet.py:450-567.  sample (N,4,h,w), controlnet_cond (N,3,8h,8w) -> ([12 residuals (N,C,h',w')], mid residual)."""
    cfg = U.normalize_config(dict(cfg, down_block_types=tuple(t.replace("2D", "3D") for t in cfg["down_block_types"])))
    boc, G, eps, hd = cfg["block_out_channels"], cfg["norm_num_groups"], cfg["norm_eps"], cfg["attention_head_dim"]
    N = sample.shape[0]
    if not torch.is_tensor(timestep):
        timestep = torch.tensor([timestep], dtype=torch.float64 if isinstance(timestep, float) else torch.int64)
    elif timestep.dim() == 0:
        timestep = timestep[None]
    timestep = timestep.expand(N)
    t_emb = U.timestep_embedding(timestep, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
    emb = U._lin(sd, "time_embedding.linear_2", F.silu(U._lin(sd, "time_embedding.linear_1", t_emb)))
    x = sample.unsqueeze(2)                                       # (N,C,1,h,w): the 2-D blocks as the F=1 case
    x = U._conv_per_frame(sd, "conv_in", x)
    x = x + cond_embedding(sd, controlnet_cond).unsqueeze(2)       # :523-525
    ulp, upc = cfg["use_linear_projection"], cfg["upcast_attention"]
    skips = [x]
    for i, t in enumerate(cfg["down_block_types"]):
        p = f"down_blocks.{i}"
        for j in range(cfg["layers_per_block"]):
            x = U.resnet_block(sd, f"{p}.resnets.{j}", x, emb, G, eps)
            if t.startswith("CrossAttn"):
                x = U.transformer3d(sd, f"{p}.attentions.{j}", x, encoder_hidden_states, hd[i], G, ulp, upcast=upc)
            skips.append(x)
        if (p + ".downsamplers.0.conv.weight") in sd:
            x = U._conv_per_frame(sd, p + ".downsamplers.0.conv", x, stride=2, padding=cfg["downsample_padding"])
            skips.append(x)
    sc = cfg["mid_block_scale_factor"]
    x = U.resnet_block(sd, "mid_block.resnets.0", x, emb, G, eps, sc)
    x = U.transformer3d(sd, "mid_block.attentions.0", x, encoder_hidden_states, hd[-1], G, ulp, upcast=upc)
    x = U.resnet_block(sd, "mid_block.resnets.1", x, emb, G, eps, sc)
    down = [F.conv2d(s.squeeze(2), sd[f"controlnet_down_blocks.{k}.weight"], sd[f"controlnet_down_blocks.{k}.bias"]) * conditioning_scale
            for k, s in enumerate(skips)]                          # :548-556
    mid = F.conv2d(x.squeeze(2), sd["controlnet_mid_block.weight"], sd["controlnet_mid_block.bias"]) * conditioning_scale
    return down, mid
