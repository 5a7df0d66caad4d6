#!/usr/bin/env python3
"""Per-basic-block instruction counts and estimated VALU issue cycles of a gfx950 .s file
(cycle classes from profiles/r01_valu_issue_rates_ubench.txt)."""
import collections
import re
import sys

CYC4 = ('v_pk_', 'v_and_or', 'v_bfi', 'v_perm', 'v_lerp', 'v_cndmask', 'v_mad', 'v_bfe', 'v_lshl_or', 'v_lshl_add',
        'v_add3', 'v_or3', 'v_cmp', 'v_mov_b32_dpp', 'v_readlane', 'v_readfirstlane', 'v_alignbit', 'v_bitop3',
        'v_mul_lo', 'v_mul_hi', 'v_min_u32', 'v_max_u32', 'v_xad', 'v_add_lshl', 'v_addc', 'v_subb', 'v_add_co',
        'v_sub_co', 'v_subrev_co', 'v_min3', 'v_max3', 'v_med3')
src = open(sys.argv[1]).read()
kern = sys.argv[2] if len(sys.argv) > 2 else 'vit_pk_kernel'
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 25
body = src[src.index(kern):]
end = body.find('.end_amdhsa_kernel')
body = body[:end] if end > 0 else body
blocks, cur = [], ('entry', [])
for l in (x.strip() for x in body.split('\n')):
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append(cur)
        cur = (l.split(':')[0], [])
    elif l and not l.startswith(';') and not l.startswith('.'):
        cur[1].append(l.split()[0])
blocks.append(cur)
for name, ins in blocks:
    if len(ins) < minlen:
        continue
    c = collections.Counter(ins)
    valu = [k for k in ins if k.startswith('v_')]
    cyc = sum(8 if 'permlane' in k else 4 if k.startswith(CYC4) else 2 for k in valu)
    print('%-10s total %4d  valu %4d  est_cycles %5d  salu %3d  ds %3d  | %s' % (
        name, len(ins), len(valu), cyc, sum(v for k, v in c.items() if k.startswith('s_')),
        sum(v for k, v in c.items() if k.startswith('ds_')), ', '.join('%s:%d' % kv for kv in c.most_common(8))))
