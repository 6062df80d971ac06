"""Census of attention_fwd4_kernel's tile bodies from hipcc -S output (no GPU): per MFMA-carrying block the instruction count,
AGPR copies, scratch traffic, and the distance (in instructions + s_nop states) from the last MFMA of every inline-asm S^T
chain to the first vector instruction that reads its result -- the hazard hipcc cannot see (12 wait states needed).
  python tools/a4_census.py /tmp/af.s
  python tools/a4_census.py --check /tmp/af.s     exit status 1 when a chain's result could be read early: the hard build step of
                                                  csrc/Makefile (a chain whose reader lies beyond its basic block must have its 12
                                                  states inside the block; every block with an asm chain is looked at)"""
import re, sys, collections

def blocks_of(body):
    out, cur = [], None
    for l in body.split('\n'):
        if re.match(r'\.LBB\d+_\d+:', l):
            cur = [l.split(':')[0], []]
            out.append(cur)
        elif cur is not None and l.startswith('\t') and not l.strip().startswith(('.', ';')):
            cur[1].append(l.strip())
    return out

def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'-?v(\d+)$', tok)
    return {int(m.group(1))} if m else set()

def hazard_distances(ins):
    res = []
    for n, x in enumerate(ins):
        if not x.startswith('v_mfma'): continue
        ops = [t.strip() for t in x.split(None, 1)[1].split(',')]
        if not ops[0].startswith('v['): continue            # AGPR destination: builtin, hipcc's own hazards
        dst = regs(ops[0])
        states = 0
        for y in ins[n + 1:]:
            if y.startswith('v_mfma'):
                o2 = [t.strip() for t in y.split(None, 1)[1].split(',')]
                if regs(o2[0]) == dst: states = None; break     # chain continues
                states += 1; continue
            if y.startswith('s_nop'): states += int(y.split()[1]) + 1; continue
            toks = [t.strip() for t in y.split(None, 1)[1].split(',')] if ' ' in y else []
            if y.startswith('v_') and any(regs(t) & dst for t in toks[1:]):
                res.append((n, states, y)); break
            states += 1
        else:
            # the block ends before any reader: whoever reads it in a following block sees at least `states` states
            if states is not None: res.append((n, states, '<end of block>'))
    return res


def check(path):
    s = open(path).read()
    bad, chains = [], 0
    for tag in ('ILb0E', 'ILb1E'):
        name = '_ZN12_GLOBAL__N_121attention_fwd4_kernel%sEEv10AttnParams' % tag
        i = s.index(name + ':')
        body = s[i:s.index('.Lfunc_end', i)]
        for lab, ins in blocks_of(body):
            for at, states, reader in hazard_distances(ins):
                chains += 1
                if states < 12: bad.append((tag, lab, at, states, reader))
    for b in bad: print('attention_fwd4: XDL-write -> VALU-read hazard: %s %s MFMA #%d: %d states before `%s`' % b, file=sys.stderr)
    if chains < 16: print('attention_fwd4: only %d inline-asm MFMA chains found (expected >= 16)' % chains, file=sys.stderr)
    return 1 if bad or chains < 16 else 0

def main(path):
    s = open(path).read()
    for tag in ('ILb0E', 'ILb1E'):
        name = '_ZN12_GLOBAL__N_121attention_fwd4_kernel%sEEv10AttnParams' % tag
        i = s.index(name + ':')
        body = s[i:s.index('.Lfunc_end', i)]
        for lab, ins in blocks_of(body):
            n = sum(x.startswith('v_mfma') for x in ins)
            if n < 32: continue
            hz = hazard_distances(ins)
            print(tag, lab, 'instrs', len(ins), 'mfma', n, 'accvgpr', sum('accvgpr' in x for x in ins),
                  'scratch', sum('scratch' in x for x in ins), 'nops', sum(x.startswith('s_nop') for x in ins),
                  'asm-chain->read states', [h[1] for h in hz])
    for m in re.finditer(r'\.vgpr_spill_count:\s*(\d+)', s): print('vgpr_spill_count', m.group(1))

if __name__ == '__main__':
    if sys.argv[1] == '--check': sys.exit(check(sys.argv[2]))
    main(sys.argv[1])
