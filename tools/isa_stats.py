#!/usr/bin/env python3
"""Per-kernel ISA resources of a hipcc -save-temps .s file: VGPRs, SGPR spills, LDS, scratch, static instruction mix.
usage: python tools/isa_stats.py <file.s> [name-substring ...]   (compile with: hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps -c x.hip)"""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read()
want = sys.argv[2:]
meta = {}
for m in re.finditer(r'\.group_segment_fixed_size:\s+(\d+)(?:(?!\.group_segment_fixed_size).)*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', txt, re.S):
    meta[m.group(2)] = dict(lds=int(m.group(1)), scratch=int(m.group(3)), sgpr=int(m.group(4)), sspill=int(m.group(5)), vgpr=int(m.group(6)), vspill=int(m.group(7)))
print("%-44s %5s %5s %6s %6s %7s %7s | %6s %6s %6s %5s %5s %5s" % ("kernel", "vgpr", "sgpr", "sspill", "vspill", "lds", "scratch", "insts", "valu", "salu", "lds", "vmem", "lane"))
for name, md in meta.items():
    if want and not any(w in name for w in want):
        continue
    try:
        i = txt.index('\n' + name + ':'); j = txt.index('.Lfunc_end', i)
    except ValueError:
        continue
    cat = Counter()
    for line in txt[i:j].split('\n'):
        s = line.strip()
        if not line.startswith('\t') or not s or s[0] in '.;':
            continue
        op = s.split()[0]
        if op.startswith(('v_readlane', 'v_writelane')): cat['lane'] += 1
        elif op.startswith('v_'): cat['valu'] += 1
        elif op.startswith('s_'): cat['salu'] += 1
        elif op.startswith('ds_'): cat['lds'] += 1
        elif op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): cat['vmem'] += 1
        else: cat['other'] += 1
    short = re.sub(r'^_Z\d+', '', name)[:44]
    print("%-44s %5d %5d %6d %6d %7d %7d | %6d %6d %6d %5d %5d %5d" % (short, md['vgpr'], md['sgpr'], md['sspill'], md['vspill'], md['lds'], md['scratch'], sum(cat.values()), cat['valu'], cat['salu'], cat['lds'], cat['vmem'], cat['lane']))
