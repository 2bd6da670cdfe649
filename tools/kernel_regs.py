"""Print VGPR/AGPR and scratch use per kernel from a hipcc -save-temps .s file (gfx950)."""
import re, subprocess, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    v = re.search(r'\.amdhsa_next_free_vgpr (\d+)', body).group(1)
    a = re.search(r'\.amdhsa_accum_offset (\d+)', body)
    sc = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body).group(1)
    try:
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        dn = name
    dn = re.sub(r'\(.*', '', dn).replace('void ', '')
    print(f"{dn:80s} regs {v:>4s} accum_off {a.group(1) if a else '-':>4s} scratch {sc}")
