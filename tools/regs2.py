"""Register / scratch use of every kernel in a -save-temps .s file:  python3 tools/regs2.py gpr_amd/_build/<file>.s"""
import re, sys
t = open(sys.argv[1]).read()
for blk in re.findall(r'- \.agpr_count:.*?\.wavefront_size', t, re.S):
    g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
    print(g('name')[:70], 'vgpr', g('vgpr_count'), 'agpr', g('agpr_count'), 'spill', g('vgpr_spill_count'), 'sgpr_spill',
          g('sgpr_spill_count'), 'scratch', g('private_segment_fixed_size'), 'lds', g('group_segment_fixed_size'))
