"""dev: one character per instruction of a kernel in a -S dump (M mfma, r/w LDS, G/S global, | waitcnt, B barrier, a accvgpr, v VALU, s SALU).  python isa_seq.py file.s kernel_prefix [chars]"""
import sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2])
j = s.index('.Lfunc_end', i)
seq = []
for l in s[i:j].split('\n'):
    l = l.strip()
    if not l or l.startswith(('.', ';', '//')):
        if l.startswith('.LBB'):
            seq.append('\n' + l.split(':')[0] + ':')
        continue
    op = l.split()[0]
    c = ('M' if op.startswith('v_mfma') else 'r' if op.startswith('ds_read') else 'w' if op.startswith('ds_write') else 'G' if op.startswith(('global_load', 'buffer_load'))
         else 'S' if op.startswith(('global_store', 'buffer_store')) else 'F' if op.startswith('flat_') else 'X' if op.startswith('scratch_') else '|' if op.startswith('s_waitcnt')
         else 'B' if op.startswith('s_barrier') else 'a' if op.startswith('v_accvgpr') else 'n' if op.startswith('s_nop') else 'v' if op.startswith('v_') else 's')
    seq.append(c)
print(''.join(seq)[:int(sys.argv[3]) if len(sys.argv) > 3 else 6000])
