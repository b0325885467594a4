#!/usr/bin/env python3
"""Resource usage and instruction mix of kernels in csrc/idahip.s (`make asm`): python tools/isa_summary.py <substring> ..."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
for pat in sys.argv[2:]:
    for m in re.finditer(r'\n(_Z\w*' + re.escape(pat) + r'\w*):', s):
        name = m.group(1)
        i = m.start()
        j = s.find('.end_amdhsa_kernel', i)
        body = s[i:j]
        k = body.find('.amdhsa_kernel')
        code = body[:k]
        print(name)
        for key in ['next_free_vgpr', 'next_free_sgpr', 'group_segment_fixed_size', 'private_segment_fixed_size', 'accum_offset']:
            mm = re.search(r'\.amdhsa_' + key + r'\s+(\S+)', body)
            print('   ', key, mm.group(1) if mm else None)
        ops = Counter()
        for l in code.split('\n'):
            if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
                f = l.split()
                if f: ops[f[0]] += 1
        print('    instructions', sum(ops.values()))
        print('   ', ops.most_common(28))
