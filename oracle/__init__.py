"""CPU oracle for the Emote-hack diffusion hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch fp32 CPU restatement of the reference's algorithm for the
path named by BASELINE.json:north_star.  It is the checker for the
HIP product in `emote_hack_amd/`; it is never the thing shipped or measured:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
  * `emote_hack_amd/` never imports it (tests/test_boundary.py enforces this by grep).

Pinning: every function here is checked against golden vectors captured in the build
container from the reference's own in-tree code (`tools/oracle/gen_golden.py` ->
`tests/golden/*.safetensors`, test: `tests/test_oracle_golden.py`).  Two pieces have no
reference-owned arithmetic in tree and are "parity unpinned" (SURVEY.md section 8c):
the diffusers scheduler (restated from the DDPM/DDIM papers, `oracle/scheduler_ref.py`)
and the AppearanceEncoder's diffusers 2-D blocks (realised as the F=1 instance of the
in-tree 3-D blocks, which IS pinned).
"""
