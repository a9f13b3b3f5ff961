"""Guard of the band kernel's software-pipelined taps: python tools/isa_pipe.py file.s <name-substring>

The pipelined plane step (render_band.hip) issues LDS reads in one asm statement and waits for them in another (`s_waitcnt lgkmcnt(N)` with
N > 0), with compiler-scheduled code in between.  The compiler does not know that the destination registers of the reads are in flight: a copy
or any other use of one of them in front of its wait would read stale bytes.  This walks every kernel whose name contains the substring,
keeps the destinations of the outstanding LDS reads in issue order (LDS operations return in order: `lgkmcnt(N)` leaves the N youngest
outstanding; scalar loads share the counter and return out of order, so they can only make a wait stricter, never weaker) and reports every
instruction that reads or writes a register a read still owns.  Branch targets are walked in listing order -- sufficient for the plane loop,
whose waits and reads sit in straight-line code.  Exit status 1 on a finding."""
import re
import sys


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(body, name):
    outstanding = []  # list of register sets, oldest first
    bad = []
    for ln, raw in enumerate(body):
        line = raw.split(";")[0].strip()
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        op, _, rest = line.partition(" ")
        toks = [t.strip() for t in re.split(r"[,\s]+", rest) if t.strip()]
        if op.startswith("ds_read"):
            dst = regs_of(toks[0]) if toks else set()
            used = set().union(*[regs_of(t) for t in toks[1:]]) if len(toks) > 1 else set()
            inflight = set().union(*outstanding) if outstanding else set()
            if (dst | used) & inflight:
                bad.append((ln, raw.strip(), sorted((dst | used) & inflight)))
            outstanding.append(dst)
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                n = int(m.group(1))
                outstanding = outstanding[len(outstanding) - n:] if n else []
            continue
        if op in ("s_barrier", "s_endpgm"):
            outstanding = []
            continue
        inflight = set().union(*outstanding) if outstanding else set()
        if inflight:
            touched = set().union(*[regs_of(t) for t in toks]) if toks else set()
            if touched & inflight:
                bad.append((ln, raw.strip(), sorted(touched & inflight)))
    return bad


def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2] if len(sys.argv) > 2 else "render_band_kernel"
    names = [n for n in re.findall(r"^(_Z\w+):", s, flags=re.M) if key in n]
    rc = 0
    for name in names:
        a = s.index(name + ":")
        body = s[a:s.index(".Lfunc_end", a)].split("\n")
        bad = check(body, name)
        n_counted = sum(1 for l in body if re.search(r"s_waitcnt lgkmcnt\([1-9]\d*\)", l))
        print(f"{name}: {n_counted} counted LDS waits, {len(bad)} uses of in-flight LDS destinations")
        for ln, txt, regs in bad[:10]:
            print(f"   line {ln}: {txt}   <- v{regs}")
        rc |= bool(bad)
    return rc


if __name__ == "__main__":
    sys.exit(main())
