"""Extract one kernel's gfx950 assembly from a --save-temps .s file and summarise it:
   python3 tools/asm_kernel.py file.s 'bwd_rs_kernelILi512ELi8ELi3' [--dump out.s]
prints instruction-class counts, spill (scratch_*) sites with their line numbers, and s_waitcnt vmcnt sites."""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
src = open(path).read().split('\n')
start = next(i for i, l in enumerate(src) if re.match(r'^_Z\S*' + re.escape(pat) + r'\S*:', l))
end = next(i for i in range(start, len(src)) if '.end_amdhsa_kernel' in src[i] or src[i].startswith('.Lfunc_end'))
body = src[start:end]
if '--dump' in sys.argv:
    open(sys.argv[sys.argv.index('--dump') + 1], 'w').write('\n'.join(body))
cnt = collections.Counter()
for i, l in enumerate(body):
    t = l.strip().split(' ')[0]
    if not t or t.startswith(('.', ';', '_')) or t.endswith(':'):
        continue
    cls = 'mfma' if 'mfma' in t else t.split('_')[0] + '_' + (t.split('_')[1] if '_' in t else '')
    cnt[cls] += 1
    if t.startswith('scratch_'):
        print('%6d  %s' % (i, l.strip()))
print(len(body), 'lines;', ', '.join('%s %d' % kv for kv in cnt.most_common(14)))
