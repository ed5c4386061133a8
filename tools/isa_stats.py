"""Instruction statistics of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only): counts of fp64 FMA /
other vector / scalar-load / LDS / lane-spill instructions per kernel plus its register use.  Usage:
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -S --cuda-device-only -o /tmp/k.s ndr_amd/csrc/kernels_mg.hip
    python tools/isa_stats.py /tmp/k.s gs_rows mf1"""
import re
import sys

txt = open(sys.argv[1]).read()
pats = sys.argv[2:]
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pats and not any(p in name for p in pats):
        continue
    cnt = lambda p: len(re.findall(p, body, re.M))
    vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', body)
    sg = re.search(r'\.amdhsa_next_free_sgpr (\d+)', body)
    lds = re.search(r'\.amdhsa_group_segment_fixed_size (\d+)', body)
    scr = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', body)
    print("%s\n   fma64 %d  mul64 %d  add64 %d  valu %d  salu %d  s_load %d  global_load %d  global_store %d  ds_read %d  ds_write %d  readlane %d  writelane %d  waitcnt %d | vgpr %s sgpr %s lds %s scratch %s"
          % (name, cnt(r"^\s+v_fmac?_f64"), cnt(r'^\s+v_mul_f64'), cnt(r'^\s+v_add_f64'), cnt(r'^\s+v_'), cnt(r'^\s+s_(?!load|waitcnt|nop|barrier)'),
             cnt(r'^\s+s_load_'), cnt(r'^\s+global_load'), cnt(r'^\s+global_store'), cnt(r'^\s+ds_read'), cnt(r'^\s+ds_write'),
             cnt(r'^\s+v_readlane'), cnt(r'^\s+v_writelane'), cnt(r'^\s+s_waitcnt'), vg and vg.group(1), sg and sg.group(1),
             lds and lds.group(1), scr and scr.group(1)))
