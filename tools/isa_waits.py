import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"^(_ZN12_GLOBAL__N_1\d+(%s)\S*):" % sys.argv[2], txt, re.M):
    name = m.group(1)
    i = m.start(); j = txt.index(".Lfunc_end", i)
    lines = [l.split(";")[0].strip() for l in txt[i:j].splitlines()]
    lines = [l for l in lines if l and (not l.startswith(".") or l.startswith(".LBB"))]
    # the main loop = from the last s_barrier-free prologue... take everything after the first "s_sleep" minus 200 as "loop"
    first_sleep = next((k for k, l in enumerate(lines) if l.startswith("s_sleep")), 0)
    print(name[:84], len(lines))
    last_mem = []
    for k, l in enumerate(lines):
        if re.match(r"(global_load|global_store|buffer_load|buffer_store)", l):
            last_mem.append((k, l[:44]))
        if l.startswith("s_waitcnt vmcnt") and k > first_sleep - 600:
            prev = last_mem[-1] if last_mem else None
            print("   %5d %-22s  last mem op %s (%d instrs before)" % (k, l, prev[1] if prev else None, k - prev[0] if prev else -1))
