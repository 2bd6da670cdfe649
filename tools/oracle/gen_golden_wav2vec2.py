#!/usr/bin/env python3
"""Generate tests/golden/wav2vec2.{safetensors,json} from the THIRD-PARTY implementation the reference calls (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/oracle/gen_golden_wav2vec2.py

/root/reference/Net.py:607-648 loads transformers' `Wav2Vec2Model` ('facebook/wav2vec2-base-960h'; no network here, so random-init
models of the same class) and reads `.last_hidden_state`.  This script instantiates that class from `transformers` (installed in the
build container, version recorded in the .json), loads name-keyed synthetic weights (emote_hack_amd.wav2vec2.wav2vec2_synth_state_dict -
both sides regenerate them, nothing is committed), feeds seeded waveforms and stores ONLY the outputs:
  tiny/*   a 2-layer, 64-wide model of the same family on 0.25 s of audio (fast CPU check of oracle/wav2vec2_ref.py)
  base/*   the wav2vec2-base configuration (94 M parameters) on 1 s of audio -> (1, 49, 768)
  base/features  the whole front-end: Wav2Vec2FeatureExtractor normalisation (the processor's) -> model -> the reference's OWN
                 windowing statements (AST-extracted from Net.py:649-667, as tools/oracle/gen_golden.py does for audio_windows)
"""
import ast
import json
import os
import sys
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import torch  # noqa: E402
import transformers  # noqa: E402
from safetensors.torch import save_file  # noqa: E402
from transformers import Wav2Vec2Config, Wav2Vec2FeatureExtractor, Wav2Vec2Model  # noqa: E402

from emote_hack_amd.synth import seeded_randn  # noqa: E402
from emote_hack_amd.wav2vec2 import wav2vec2_synth_state_dict  # noqa: E402
from tests import cases  # noqa: E402

torch.set_grad_enabled(False)


def build(cfg):
    m = Wav2Vec2Model(Wav2Vec2Config(**cfg)).eval()
    sd = wav2vec2_synth_state_dict(cfg)
    own = m.state_dict()
    assert set(sd) | {"masked_spec_embed"} >= set(own), sorted(set(own) - set(sd))[:5]
    m.load_state_dict({k: sd.get(k, own[k]) for k in own}, strict=True)
    return m


def reference_windows(hidden_states, m, n):
    tree = ast.parse(open("/root/reference/Net.py").read())
    cls = next(x for x in tree.body if isinstance(x, ast.ClassDef) and x.name == "Wav2VecFeatureExtractor")
    fn = next(x for x in cls.body if isinstance(x, ast.FunctionDef) and x.name == "extract_features_from_wav")
    start = next(i for i, st in enumerate(fn.body) if isinstance(st, ast.Assign) and getattr(st.targets[0], "id", "") == "num_frames")
    stop = next(i for i, st in enumerate(fn.body) if isinstance(st, ast.Assign) and getattr(st.targets[0], "id", "") == "all_features"
                and isinstance(st.value, ast.Call) and getattr(st.value.func, "attr", "") == "stack")
    ns = dict(torch=torch, hidden_states=hidden_states, m=m, n=n, self=SimpleNamespace(device="cpu"))
    exec(compile(ast.Module(body=fn.body[start:stop + 1], type_ignores=[]), "Net.py", "exec"), ns)
    return ns["all_features"]


def main():
    T = {}
    tiny = build(cases.WAV2VEC2_TINY)
    x = 0.5 * seeded_randn((1, 4000), 501)
    T["tiny/out"] = tiny(x).last_hidden_state.contiguous()
    base = build({})
    wave = 0.1 * seeded_randn((16000,), 502) + 0.05 * torch.sin(torch.arange(16000) * 0.05)
    proc = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True, return_attention_mask=False)
    iv = proc(wave.numpy(), sampling_rate=16000, return_tensors="pt").input_values          # Net.py:639
    hs = base(iv).last_hidden_state                                                          # Net.py:643-644
    T["base/input_values"] = iv.contiguous()
    T["base/out"] = hs.contiguous()
    T["base/features"] = reference_windows(hs, 2, 2).contiguous()                            # Net.py:646-667
    save_file(T, os.path.join(cases.GOLDEN_DIR, "wav2vec2.safetensors"))
    json.dump({"transformers": transformers.__version__, "torch": torch.__version__,
               "shapes": {k: list(v.shape) for k, v in T.items()}}, open(os.path.join(cases.GOLDEN_DIR, "wav2vec2.json"), "w"), indent=1)
    print({k: tuple(v.shape) for k, v in T.items()}, "transformers", transformers.__version__)


if __name__ == "__main__":
    main()
