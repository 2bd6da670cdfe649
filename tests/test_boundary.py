"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/emo_hip.h
declares (no compute calls without a GPU), the product never imports the oracle, the host-side mirror has
the reference's interface (ctor kwargs, state-dict keys, error behaviour)."""
import ctypes
import json
import os
import re
import subprocess

import pytest
import torch

from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "emo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(emo_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from emote_hack_amd import _lib
    lib = _lib.load()
    declared = header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/emo_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, set(_lib.SIGNATURES) ^ set(declared)
    assert lib.emo_version() >= 100
    # extern "C": no mangled emo_ entry points
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(declared) <= exported


def test_ctypes_struct_layout_matches_header():
    """Field order of the params structs must match the header (a stale .so once silently mis-read a field)."""
    from emote_hack_amd._lib import AttentionParams, GemmParams
    src = open(os.path.join(ROOT, "include", "emo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for cname, struct in (("emo_gemm_params", GemmParams), ("emo_attention_params", AttentionParams)):
        body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} " + cname, src, flags=re.S).group(1)
        fields = re.findall(r"(?:const\s+)?(?:void|float|int64_t|uint32_t|int32_t|int)\s*\*?\s*([A-Za-z_0-9]+)\s*;", body)
        assert fields == [f[0] for f in struct._fields_], (cname, fields)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "emote_hack_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"
                assert "/root/reference" not in txt, f"{f} reads the reference at run time"


def test_no_cpu_fallback():
    """The product path must fail loudly without a HIP device."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from emote_hack_amd import ops
    from emote_hack_amd._lib import EmoHipError
    from emote_hack_amd.unet import UNet3DConditionModel
    with pytest.raises(EmoHipError):
        ops.silu(torch.zeros(4))
    m = UNet3DConditionModel(**cases.TINY)
    with pytest.raises(EmoHipError):
        m(torch.zeros(1, 4, 1, 16, 16), 1, torch.zeros(1, 5, 32))


# ------------------------------------------------------------------ UNet surface
@pytest.fixture(scope="module")
def ints():
    return json.load(open(os.path.join(cases.GOLDEN_DIR, "ints.json")))


def test_state_dict_keys_match_reference(ints):
    import hashlib
    from emote_hack_amd.spec import build_spec, param_shapes
    for cfg, key in ((cases.TINY_MOTION, "tiny_motion_keys"), (cases.TINY_LINEAR, "tiny_linear_keys")):
        mine = {k: list(v) for k, v in param_shapes(build_spec(cfg)).items()}
        assert mine == ints[key]
    for cfg, key in ((cases.SD15_MOTION, "sd15_motion"), (dict(cases.SD15_MOTION, motion_module_mid_block=True), "sd15_motion_mid"),
                     (cases.SD15, "sd15")):
        shp = param_shapes(build_spec(cfg))
        canon = "\n".join(f"{k}:{','.join(map(str, shp[k]))}" for k in sorted(shp))
        d = ints[key + "_digest"]
        assert len(shp) == d["n_keys"]
        assert hashlib.sha256(canon.encode()).hexdigest() == d["sha256"]
        n_params = sum(int(torch.Size(s).numel()) for k, s in shp.items() if not k.endswith("pos_encoder.pe"))
        assert n_params == d["n_params"]


def test_appearance_encoder_key_set_is_the_gutted_reference_class(ints):
    """A15: the ReferenceNet of the reference replaces nine sub-modules of up_blocks[3].attentions[2] by parameter-less
    ones (appearance_encoder.py:613-621; golden = those assignments read from the reference ctor by AST).  The product class
    must own exactly the F=1 no-motion UNet's keys (conv_norm_out / conv_out dropped) minus everything under a replaced
    path - a reference-shaped checkpoint then loads strictly - and only norm / proj_in / norm1 may remain in that block."""
    from emote_hack_amd.appearance_encoder import AppearanceEncoderModel
    from emote_hack_amd.spec import build_spec, param_shapes
    repl = ints["appearance_encoder_gutted"]
    assert len(repl) == 9 and all(n == 0 for _, _, n in repl)
    for cfg in (cases.SD15, cases.TINY):
        full = param_shapes(build_spec(cfg, has_out=False))
        want = {k: tuple(v) for k, v in full.items() if not any(k == p or k.startswith(p + ".") for p, _, _ in repl)}
        m = AppearanceEncoderModel(**cfg)
        got = {k: tuple(v) for k, v in m._shapes.items()}
        assert got == want
        blk = sorted(k for k in got if k.startswith("up_blocks.3.attentions.2."))
        assert blk == sorted("up_blocks.3.attentions.2." + n for n in (
            "norm.weight", "norm.bias", "proj_in.weight", "proj_in.bias",
            "transformer_blocks.0.norm1.weight", "transformer_blocks.0.norm1.bias"))
        sd = {k: torch.zeros(s) for k, s in want.items()}
        assert m.load_state_dict(sd, strict=True) == ([], [])
        with pytest.raises(RuntimeError, match="unexpected"):   # a full-UNet checkpoint is NOT a ReferenceNet checkpoint
            m.load_state_dict({k: torch.zeros(s) for k, s in full.items()}, strict=True)
        # the last bank of the pairing order is the gutted block's LN1 output
        assert "up_blocks.3.attentions.2" in m.bank_order("midup")


def test_load_state_dict_merges_partial_loads():
    """The reference loads the 2-D checkpoint and then the motion-module checkpoint, both strict=False
    (unet_controlnet.py:485-525, animation.py:116-135): `missing` is relative to each incoming dict, the model becomes usable
    once the MERGED master is complete, and an incomplete model names its absent keys instead of dying in a KeyError."""
    from emote_hack_amd._lib import EmoHipError
    from emote_hack_amd.unet import UNet3DConditionModel
    m = UNet3DConditionModel(**cases.TINY_MOTION)
    sd = {k: torch.zeros(s) for k, s in m._shapes.items()}
    base = {k: v for k, v in sd.items() if "motion_modules" not in k}
    mm = {"module." + k: v for k, v in sd.items() if "motion_modules" in k}
    missing, unexpected = m.load_state_dict(base, strict=False)
    assert missing and all("motion_modules" in k for k in missing) and not unexpected
    assert len(m._absent_keys()) == len(missing)
    with pytest.raises(EmoHipError, match="no value yet"):
        m._begin(torch.zeros(1, 4, 1, 16, 16), 1, torch.zeros(1, 5, 32))
    stripped = {k[len("module."):]: v for k, v in mm.items()}       # animation.py:126-133 strips the prefixes
    missing2, _ = m.load_state_dict(stripped, strict=False)
    assert set(missing2) == set(base) and m._absent_keys() == []


def test_vae_key_listing_is_the_published_sd_vae():
    """(f)2: the AutoencoderKL of SD-1.x has 248 tensors / 83,653,863 parameters (the published `vae/` checkpoint of
    runwayml/stable-diffusion-v1-5 and stabilityai/sd-vae-ft-*); the product's key walk reproduces both and the encoder /
    decoder / quant split.  (Parity of the arithmetic is unpinned: diffusers is absent.)"""
    import math
    from emote_hack_amd.vae import VAE_DEFAULTS, vae_param_shapes
    d = vae_param_shapes(VAE_DEFAULTS)
    assert len(d) == 248 and sum(math.prod(s) for s in d.values()) == 83653863
    assert d["decoder.mid_block.attentions.0.to_q.weight"] == (512, 512) and d["encoder.conv_out.weight"] == (8, 512, 3, 3)
    assert d["quant_conv.weight"] == (8, 8, 1, 1) and d["post_quant_conv.weight"] == (4, 4, 1, 1)
    assert "decoder.up_blocks.2.resnets.0.conv_shortcut.weight" in d and "decoder.up_blocks.3.upsamplers.0.conv.weight" not in d


def test_bank_pairing_order(ints):
    from emote_hack_amd.spec import build_spec, reference_block_order
    strip = lambda names: [n.replace(".transformer_blocks.0", "") for n in names]
    assert reference_block_order(build_spec(cases.TINY_MOTION), "midup") == strip(ints["bank_order_tiny_midup"])
    assert reference_block_order(build_spec(cases.TINY_MOTION), "full") == strip(ints["bank_order_tiny_full"])
    assert reference_block_order(build_spec(cases.SD15_MOTION), "midup") == strip(ints["bank_order_sd15_midup"])


def test_unet_ctor_surface_and_errors():
    from emote_hack_amd.unet import UNet3DConditionModel
    m = UNet3DConditionModel(**cases.SD15_MOTION)
    assert m.config.sample_size == 64 and m.in_channels == 4 and m.config.attention_head_dim == (8, 8, 8, 8)
    assert m.config["norm_eps"] == 1e-5
    with pytest.raises(TypeError):
        UNet3DConditionModel(not_a_kwarg=1)
    with pytest.raises(ValueError, match="does not exist"):   # 2-D names must be renamed (unet_3d_blocks.py:103)
        UNet3DConditionModel(**dict(cases.SD15, down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)))
    with pytest.raises(ValueError, match="unknown mid_block_type"):
        UNet3DConditionModel(**dict(cases.SD15, mid_block_type="Foo"))
    sd = {k: torch.zeros(s) for k, s in m._shapes.items()}
    del sd["conv_in.weight"]
    with pytest.raises(RuntimeError):
        m.load_state_dict(sd, strict=True)
    missing, unexpected = m.load_state_dict(dict(sd, extra=torch.zeros(1)), strict=False)
    assert missing == ["conv_in.weight"] and unexpected == ["extra"]


def test_unet_config_yaml_rules(tmp_path):
    """2D->3D names, unknown keys dropped, norm_num_groups honoured, '1e-05' cast to float."""
    from emote_hack_amd.config import unet_config_from_yaml
    y = tmp_path / "unet-config.yaml"
    y.write_text("""
denoising_unet_config:
  default:
    act_fn: silu
    attention_head_dim: 8
    block_out_channels: [320, 640, 1280, 1280]
    cross_attention_dim: 768
    down_block_types: ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"]
    up_block_types: ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"]
    norm_eps: 1e-05
    norm_num_groups: 4
    sample_size: 64
    projection_class_embeddings_input_dim: null
""")
    cfg = unet_config_from_yaml(str(y))
    assert cfg["down_block_types"][0] == "CrossAttnDownBlock3D" and cfg["up_block_types"][0] == "UpBlock3D"
    assert cfg["norm_eps"] == 1e-5 and isinstance(cfg["norm_eps"], float)
    assert cfg["norm_num_groups"] == 4 and "projection_class_embeddings_input_dim" not in cfg


# ------------------------------------------------------------------ host-side INT logic, product vs goldens / oracle
def test_context_windows_bit_exact(ints):
    from emote_hack_amd.context import uniform
    for c in ints["windows"]:
        f, ctx, stride, ov = c["args"]
        assert list(uniform(0, 50, f, ctx, stride, ov)) == c["windows"]
    for c in ints["windows_step"]:
        f, ctx, stride, ov = c["args"]
        assert list(uniform(c["step"], 50, f, ctx, stride, ov)) == c["windows"]


def test_scheduler_tables_golden(ints):
    """A3 regression pin: INT timestep tables bit-exact, update coefficients to 1e-12 (tests/golden/ints.json
    "scheduler_tables", generated from oracle/scheduler_ref.py - parity UNPINNED against diffusers, which is absent)."""
    from emote_hack_amd import DDIMScheduler, DDPMScheduler
    from emote_hack_amd.pipeline import EMOAnimationPipeline
    assert {"ddim_50", "ddpm_50", "ddpm0_50"} <= set(ints["scheduler_tables"])
    assert ints["scheduler_tables"]["ddpm_50"]["timesteps"] == list(range(981, 0, -20))      # what the pipeline runs (:105-117)
    for name, tab in ints["scheduler_tables"].items():
        kind, n = name.split("_")
        sch = DDIMScheduler() if kind == "ddim" else DDPMScheduler()
        if kind != "ddpm0":      # through the pipeline ctor, which forces steps_offset = 1 on every scheduler that has the key
            EMOAnimationPipeline(unet=type("U", (), {"device": "cpu"})(), scheduler=sch)
            assert sch.config.steps_offset == 1
        assert sch.set_timesteps(int(n)) == tab["timesteps"]
        for t, want in zip(tab["timesteps"], tab["coefficients"]):
            got = sch.coefficients(t)
            assert all(abs(a - b) <= 1e-12 * max(1.0, abs(b)) for a, b in zip(got, want)), (name, t, got, want)


def test_ddim_step_inverts_the_in_tree_next_step():
    """The only scheduler algebra the reference holds in tree is the DDIM inversion `next_step`
    (EMOAnimationPipeline.py:379-400): alpha_t from `timestep - T // n` (final_alpha_cumprod below 0), pred_x0 =
    (x - sqrt(1-a) eps) / sqrt(a), x_next = sqrt(a_next) pred_x0 + sqrt(1 - a_next) eps.  Restated here over the PRODUCT
    scheduler object (same attribute names as diffusers': alphas_cumprod, final_alpha_cumprod, config.num_train_timesteps,
    num_inference_steps); it must undo the product's eta=0 DDIM update with the same eps at every step of the table, and its
    pred_x0 must be the x0 the forward process started from."""
    from emote_hack_amd import DDIMScheduler
    from emote_hack_amd.synth import seeded_randn
    sch = DDIMScheduler()
    ts = sch.set_timesteps(50)

    def next_step(model_output, timestep, x):                       # :379-400
        nxt = timestep
        timestep = min(timestep - sch.config.num_train_timesteps // sch.num_inference_steps, 999)
        a_t = float(sch.alphas_cumprod[timestep]) if timestep >= 0 else float(sch.final_alpha_cumprod)
        a_next = float(sch.alphas_cumprod[nxt])
        pred_x0 = (x - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        return a_next ** 0.5 * pred_x0 + (1 - a_next) ** 0.5 * model_output, pred_x0

    x0, eps = seeded_randn((1, 4, 2, 8, 8), 1).double(), seeded_randn((1, 4, 2, 8, 8), 2).double()
    for t in ts:
        a_t = float(sch.alphas_cumprod[t])
        x_t = a_t ** 0.5 * x0 + (1 - a_t) ** 0.5 * eps
        c_x, c_eps, c_n = sch.coefficients(t, 0.0)
        assert c_n == 0.0
        x_prev = c_x * x_t + c_eps * eps                             # what emo_cfg_step applies (:817)
        back, pred_x0 = next_step(eps, t, x_prev)
        torch.testing.assert_close(back, x_t, rtol=1e-9, atol=1e-9)
        torch.testing.assert_close(pred_x0, x0, rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("kind", ["ddim", "ddpm"])
def test_scheduler_matches_oracle(kind):
    from emote_hack_amd import DDIMScheduler, DDPMScheduler
    from oracle.scheduler_ref import SchedulerRef
    mine = DDIMScheduler() if kind == "ddim" else DDPMScheduler(steps_offset=1)
    ref = SchedulerRef(kind)
    assert mine.set_timesteps(50) == ref.set_timesteps(50)          # INT, bit-exact
    assert mine.timesteps == list(range(981, 0, -20))               # the offset the pipeline ctor forces on both (:105-117)
    assert DDPMScheduler().set_timesteps(50) == SchedulerRef("ddpm", steps_offset=0).set_timesteps(50) == list(range(980, -1, -20))
    with pytest.raises(NotImplementedError):
        DDIMScheduler(timestep_spacing="trailing")
    for t in mine.timesteps:
        a, b = mine.coefficients(t), ref.coefficients(t)
        assert all(abs(x - y) <= 1e-12 * max(1, abs(y)) for x, y in zip(a, b)), (t, a, b)
    mine2 = DDIMScheduler(beta_schedule="scaled_linear")
    ref2 = SchedulerRef("ddim", beta_schedule="scaled_linear")
    mine2.set_timesteps(25); ref2.set_timesteps(25)
    assert mine2.coefficients(41)[0] == pytest.approx(ref2.coefficients(41)[0], rel=1e-12)
    with pytest.raises(ValueError):
        DDIMScheduler(clip_sample=True)


def test_attention_argument_checks_run_before_any_launch():
    """emo_attention validates its geometry on the host (no GPU needed: the pointers are never dereferenced there): null pointers,
    a head dim that is not a multiple of the 16-byte vector, and - since the K / V^T tiles are addressed through 32-bit buffer offsets
    (csrc/attention.hip loader) - a (batch row, head) slab beyond 1 GB are refused with an error string, never launched."""
    import ctypes as C
    from emote_hack_amd import _lib
    lib = _lib.load()
    fake = 0x1000          # any non-null address: argument checks only
    def params(**kw):
        p = _lib.AttentionParams()
        base = dict(q=fake, ldq=320, k0=fake, ldk0=640, v0t=fake, ldv0t=4096, Lk0=4096, k1=None, ldk1=0, v1t=None, ldv1t=0, Lk1=0, seg0_div=1,
                    seg1_div=1, seg1_first_batch=0, seg1_skip=0, out=fake, ldo=320, B=2, Lq=4096, heads=8, d=40, scale=0.158, dtype=_lib.EMO_BF16,
                    seg1_row=None)
        base.update(kw)
        for k, v in base.items():
            setattr(p, k, v)
        return p
    rc = lib.emo_attention(C.byref(params(q=None)), None)
    assert rc == -5 and b"null" in lib.emo_last_error_string()
    rc = lib.emo_attention(C.byref(params(d=44)), None)
    assert rc == -1 and b"head dim" in lib.emo_last_error_string()
    rc = lib.emo_attention(C.byref(params(Lk0=1 << 20, ldk0=640, ldv0t=1 << 20)), None)       # 2^20 keys x 1280 B rows = 1.3 GB of K per batch row
    assert rc == -1 and b"exceeds 1 GB" in lib.emo_last_error_string()
    rc = lib.emo_attention(C.byref(params(k1=fake, v1t=fake, Lk1=1 << 20, ldk1=640, ldv1t=1 << 20, seg1_div=2)), None)
    assert rc == -1 and b"segment 1" in lib.emo_last_error_string()
