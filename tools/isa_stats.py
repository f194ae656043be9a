#!/usr/bin/env python
"""developer tool: registers, scratch and LDS of every kernel in a hipcc -save-temps device .s file
   hipcc ... -c x.hip -o /tmp/x.o -save-temps=obj ; python tools/isa_stats.py /tmp/x-hip-amdgcn-amd-amdhsa-gfx950.s [filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, flags=re.S):
    blk = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if flt and flt not in name: continue
    g = lambda k: re.search(rf"\.{k}:\s+(\d+)", blk).group(1)
    print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} vspill {g('vgpr_spill_count'):>3s} sspill {g('sgpr_spill_count'):>3s}")
