"""Build-time check of the GEMM kernels' ISA.

gather_gemm_k and wgrad_gemm_k issue their LDS fragment reads (ds_read_b128 / ds_read_b64_tr_b16) as inline
asm so that the compiler's wait-count pass neither drains the LDS-DMA ring nor the LDS queue in front of
them; the price is that the compiler no longer knows those registers are filled asynchronously.  The source
guards every group with a tied s_waitcnt; this check proves it on the generated code: on every path of the
kernel's control-flow graph, between such a read and the wait that covers it no instruction may touch a
register the read is still filling (the hardware does not interlock them)."""
import re

_BR = re.compile(r"^(s_cbranch_\w+|s_branch)\s+(\S+)")


def _blocks(body):
    """Split a kernel's text into basic blocks: {label: (instructions, successors)}; entry label is '^'."""
    blocks, order, cur, name = {}, [], [], "^"
    for raw in body.splitlines()[1:]:
        ln = raw.split(";")[0].strip()
        if not ln:
            continue
        if ln.endswith(":") and not ln.startswith("s_") and " " not in ln:
            blocks[name] = cur; order.append(name)
            name, cur = ln[:-1], []
            continue
        if ln.startswith("."):
            continue
        cur.append(ln)
        if _BR.match(ln) or ln.startswith("s_endpgm") or ln.startswith("s_setpc"):
            blocks[name] = cur; order.append(name)
            name, cur = f"{name}+{len(order)}", []
    blocks[name] = cur; order.append(name)
    succ = {}
    for i, n in enumerate(order):
        ins = blocks[n]
        nxt = order[i + 1] if i + 1 < len(order) else None
        last = ins[-1] if ins else ""
        m = _BR.match(last)
        if last.startswith("s_endpgm"):
            succ[n] = []
        elif m and m.group(1) == "s_branch":
            succ[n] = [m.group(2)]
        elif m:
            succ[n] = [m.group(2)] + ([nxt] if nxt else [])
        else:
            succ[n] = [nxt] if nxt else []
    return blocks, succ


def _regs(text):
    used = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        used |= set(range(int(a), int(b) + 1))
    used |= set(int(x) for x in re.findall(r"\bv(\d+)\b", text))
    return used


def _mark_asm(body):
    """kernel text with ' @asm' appended to the instructions between ;;#ASMSTART and ;;#ASMEND"""
    marked, inasm = [], False
    for raw in body.splitlines():
        if "#ASMSTART" in raw:
            inasm = True
            continue
        if "#ASMEND" in raw:
            inasm = False
            continue
        code = raw.split(";")[0].rstrip()
        marked.append(code + " @asm" if (inasm and code.strip()) else raw)
    return "\n".join(marked)


def tr_asm_hazards(asm_text, kernel_prefix="_ZN5rbvae12wgrad_gemm_k", read_ops=("ds_read_b64_tr_b16",)):
    """Returns a list of 'kernel: message' strings (empty = clean).  Forward data flow over the kernel's CFG;
    the state is the in-order queue of in-flight LDS operations, one visit per (block, state).  Only the reads issued
    as inline asm carry registers (the compiler waits for its own reads itself); every other LDS operation -- the
    compiler's reads and writes -- takes a register-less place in the queue, so a compiler wait such as lgkmcnt(4)
    behind five of its own reads is read for what it covers."""
    out = []
    for m in re.finditer(r"^(" + re.escape(kernel_prefix) + r"\w+):.*?s_endpgm", asm_text, re.S | re.M):
        blocks, succ = _blocks(_mark_asm(m.group(0)))
        seen, work, msgs = set(), [("^", ())], set()
        while work:
            name, state = work.pop()
            if (name, state) in seen or name not in blocks:
                continue
            seen.add((name, state))
            if len(seen) > 200000:
                msgs.add("state explosion: check aborted"); break
            reads = list(state)
            for ln in blocks[name]:
                asm = ln.endswith("@asm")
                ln = ln.replace(" @asm", "")
                op, _, rest = ln.partition(" ")
                if op in read_ops and asm:
                    dst, _, addr = rest.partition(",")
                    pending = set().union(*reads) if reads else set()
                    if _regs(addr) & pending:
                        msgs.add(f"in-flight register used as address: {ln}")
                    if _regs(dst) & pending:
                        msgs.add(f"in-flight register overwritten by a second read: {ln}")
                    reads.append(frozenset(_regs(dst)))
                    reads = reads[-15:]
                    continue
                if op == "s_waitcnt" and "lgkmcnt" in rest:
                    n = int(re.search(r"lgkmcnt\((\d+)\)", rest).group(1))
                    # LDS returns in order: at most the n youngest operations are still outstanding
                    reads = reads[len(reads) - n:] if n else []
                    continue
                if reads:
                    hit = _regs(rest) & set().union(*reads)
                    if hit:
                        msgs.add(f"touches in-flight v{sorted(hit)}: {ln}")
                if op.startswith("ds_"):
                    reads.append(frozenset())          # a compiler LDS operation: a place in the queue, no registers
                    reads = reads[-15:]
            for nx in succ[name]:
                work.append((nx, tuple(reads)))
        out += [f"{m.group(1)}: {x}" for x in sorted(msgs)]
    return out


_VM_OPS = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "scratch_load",
           "scratch_store", "flat_load", "flat_store")


def asm_vmem_load_hazards(asm_text, kernel_prefix, load_op="global_load_dwordx4"):
    """The same proof for register-destination loads issued as inline asm (conv_halo_k's patch pieces): between such a
    load (inside ;;#ASMSTART / ;;#ASMEND) and the s_waitcnt vmcnt(N) that covers it, nothing may touch its destination.
    vmcnt counts every vector-memory operation in issue order (LDS-DMA and stores included): the queue holds them all,
    only the asm loads carry registers."""
    out = []
    for m in re.finditer(r"^(" + re.escape(kernel_prefix) + r"\w+):.*?s_endpgm", asm_text, re.S | re.M):
        blocks, succ = _blocks(_mark_asm(m.group(0)))
        seen, work, msgs = set(), [("^", ())], set()
        while work:
            name, state = work.pop()
            if (name, state) in seen or name not in blocks:
                continue
            seen.add((name, state))
            if len(seen) > 200000:
                msgs.add("state explosion: check aborted"); break
            q = list(state)                       # in-order queue: frozenset of destination registers (empty = no registers)
            for ln in blocks[name]:
                asm = ln.endswith("@asm")
                ln = ln.replace(" @asm", "")
                op, _, rest = ln.partition(" ")
                if op == "s_waitcnt" and "vmcnt" in rest:
                    n = int(re.search(r"vmcnt\((\d+)\)", rest).group(1))
                    q = q[len(q) - n:] if n else []
                    continue
                pending = set().union(*q) if q else set()
                if pending and _regs(rest) & pending and not (asm and op == load_op and not (_regs(rest.partition(",")[0]) & pending)):
                    hit = _regs(rest) & pending
                    msgs.add(f"touches in-flight v{sorted(hit)}: {ln}")
                if any(op.startswith(v) for v in _VM_OPS):
                    q.append(frozenset(_regs(rest.partition(",")[0])) if (asm and op == load_op) else frozenset())
                    q = q[-63:]
            for nx in succ[name]:
                work.append((nx, tuple(q)))
        out += [f"{m.group(1)}: {x}" for x in sorted(msgs)]
    return out
