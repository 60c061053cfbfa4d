#!/usr/bin/env python3
"""Static census of a gfx950 code object: per function (and per backward-branch loop body) the number of
multiplies, other VALU instructions, private-segment (scratch) loads / stores, LDS accesses and waits.

    tools/isa_census.py <object.o | code-object> [function-substring ...]

Used to track what the register allocator spills in the BLS lane kernels without a GPU visit: the private
segment traffic of a kernel is (scratch instructions x dwords) per trip, weighted by hand with the trip counts.
"""
import os, re, subprocess, sys, tempfile, collections

LLVM = "/opt/rocm/lib/llvm/bin"


def device_object(path):
    with open(path, "rb") as f:
        head = f.read(64)
    if b"__CLANG_OFFLOAD_BUNDLE__" not in open(path, "rb").read(4096) and head[18:20] == b"\xe0\x00":
        return path
    d = tempfile.mkdtemp(prefix="isa_census_")
    out = os.path.join(d, "dev.co")
    r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + path,
                        "--targets=hip-amdgcn-amd-amdhsa--gfx950", "--output=" + out], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
        # host object with an embedded fat binary: let objdump extract it next to a copy
        cp = os.path.join(d, os.path.basename(path))
        subprocess.run(["cp", path, cp], check=True)
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", cp], capture_output=True, text=True)
        cands = [os.path.join(d, f) for f in os.listdir(d) if "amdgcn" in f]
        if not cands:
            raise SystemExit("no gfx950 code object in " + path)
        out = cands[0]
    return out


def dwords(op):
    m = re.search(r"dwordx(\d)", op)
    if m:
        return int(m.group(1))
    m = re.search(r"_b(\d+)$", op)
    if m:
        return max(1, int(m.group(1)) // 32)
    return 1


def census(lines):
    c = collections.Counter()
    for addr, op, rest in lines:
        c["insn"] += 1
        if op.startswith("v_mad_u64_u32") or op.startswith("v_mul_lo_u32") or op.startswith("v_mul_hi_u32"):
            c["mul"] += 1
        elif op.startswith("v_accvgpr"):
            c["acc_mov"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("scratch_load"):
            c["sc_ld"] += 1
            c["sc_ld_dw"] += dwords(op)
        elif op.startswith("scratch_store"):
            c["sc_st"] += 1
            c["sc_st_dw"] += dwords(op)
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
            c["glob"] += 1
        elif op == "s_waitcnt":
            c["wait"] += 1
            if "vmcnt" in rest:
                c["wait_vm"] += 1
        elif op.startswith("s_swappc") or op.startswith("s_setpc"):
            c["call"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return c


def fmt(c):
    return ("insn %7d  mul %7d  valu %7d  accmov %6d  scratch ld %5d (%6d dw) st %5d (%6d dw)  lds %5d  glob %4d  wait %5d (vm %5d)  call %3d"
            % (c["insn"], c["mul"], c["valu"], c["acc_mov"], c["sc_ld"], c["sc_ld_dw"], c["sc_st"], c["sc_st_dw"], c["lds"], c["glob"],
               c["wait"], c["wait_vm"], c["call"]))


def long_branch_clobbers(obj_path):
    """Functions in which a long-branch expansion (s_getpc_b64 / s_add_u32 / s_addc_u32 / s_setpc_b64 through one SGPR pair)
    uses s[30:31] -- the return address -- without the function restoring it afterwards: such a function returns into its own
    loop and never terminates (LLVM AMDGPU branch relaxation in a function without calls; csrc/common.h
    ECG_LONG_BRANCH_GUARD).  Returns [(function, address)]."""
    try:
        obj = device_object(obj_path)
    except SystemExit:
        return []  # a host-only object
    txt = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", obj], capture_output=True, text=True).stdout
    bad = []
    cur, pending, is_kernel = None, None, False
    for ln in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", ln)
        if m:
            # (a KERNEL has no return address: it ends in s_endpgm, and s[30:31] is an ordinary pair the scavenger may hand out --
            # round 6: k_h2c_map_row, 214 KB with calls in it, has such a branch)
            if pending and not is_kernel:
                bad.append(pending)
            cur, pending, is_kernel = m.group(2), None, False
            continue
        if re.search(r"\bs_endpgm\b", ln):
            is_kernel = True
        if "s_getpc_b64 s[30:31]" in ln:
            a = re.search(r"//\s*([0-9A-Fa-f]+):", ln)
            pending = (cur, a.group(1) if a else "?")
        elif pending and re.search(r"v_readlane_b32 s3[01],|s_mov_b64 s\[30:31\]|s_load_dwordx2 s\[30:31\]", ln):
            pending = None  # restored from its save slot before the return
        elif pending and "s_swappc_b64 s[30:31]" in ln:
            pending = None  # a call site: s[30:31] is rewritten anyway, so it is saved around
    if pending and not is_kernel:
        bad.append(pending)
    return bad


def main():
    if "--check-long-branches" in sys.argv:
        rc = 0
        for path in [a for a in sys.argv[1:] if not a.startswith("--")]:
            for fn, addr in long_branch_clobbers(path):
                print(f"{path}: {fn}: long branch at {addr} clobbers the return address s[30:31]")
                rc = 1
        sys.exit(rc)
    obj = device_object(sys.argv[1])
    want = sys.argv[2:]
    txt = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", obj], capture_output=True, text=True).stdout
    funcs = collections.OrderedDict()
    cur = None
    for ln in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", ln)
        if m:
            cur = m.group(2)
            funcs[cur] = []
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", ln)
        if m and cur is not None:
            funcs[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    demangle = subprocess.run(["c++filt"], input="\n".join(funcs), capture_output=True, text=True).stdout.splitlines()
    for (name, lines), dn in zip(funcs.items(), demangle):
        short = re.sub(r"\(.*", "", dn)
        if want and not any(w in short for w in want):
            continue
        if not lines:
            continue
        size = lines[-1][0] - lines[0][0] + 8
        print("%s  [%d KB]" % (short, size // 1024))
        print("    total      " + fmt(census(lines)))
        # loops: a branch whose target precedes it; report the body between target and branch
        addr_index = {a: i for i, (a, _, _) in enumerate(lines)}
        loops = []
        for i, (a, op, rest) in enumerate(lines):
            if op.startswith("s_cbranch") or op == "s_branch":
                m = re.search(r"<[^>]*\+0x([0-9a-f]+)>|(\d+)\s*$", rest)
                # objdump prints the target as an absolute address comment in newer versions: fall back to the simm16 offset
                tm = re.search(r"//|$", rest)
                off = re.match(r"(-?\d+)", rest)
                if off:
                    o = int(off.group(1))
                    if o >= 32768:
                        o -= 65536
                    tgt = a + 4 + 4 * o
                    if tgt < a and tgt in addr_index:
                        loops.append((addr_index[tgt], i))
        for (s, e) in loops:
            body = lines[s:e + 1]
            if len(body) < int(os.environ.get("CENSUS_MIN_LOOP", "200")):
                continue
            print("    loop %6x..%6x [%4d KB]  " % (lines[s][0], lines[e][0], (lines[e][0] - lines[s][0]) // 1024) + fmt(census(body)))


if __name__ == "__main__":
    main()
