"""Build libecgpu.so (gfx950 only) and the test-side helpers.

    python -m ethereum_consensus_amd.build            # product library
    python -m ethereum_consensus_amd.build --hostsim  # + tests/hostsim kernel simulator (g++)

hipcc cross-compiles for gfx950 without a GPU.  Objects are cached by source mtime so that an
edit to one .hip rebuilds one object.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# ECGPU_EXPERIMENTS=1: the library with the kernel builds and dispatch controls that lost on measurement and are the default at
# no size on any box (two-wave k_miller2 / k_finalexp2, the one-lane k_finalexp, ECGPU_PAIRING=auto1, ECGPU_SIDE_OVERLAP,
# ECGPU_M2_WAVES, ECGPU_FINALEXP_LANES) -- lib/libecgpu_exp.so next to the product, selected with ECGPU_LIB
EXPERIMENTS = os.environ.get("ECGPU_EXPERIMENTS") == "1"
OBJDIR = os.path.join(LIBDIR, "obj_exp" if EXPERIMENTS else "obj")
LIB = os.path.join(LIBDIR, "libecgpu_exp.so" if EXPERIMENTS else "libecgpu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
VM3_GEN_ARGS = os.environ.get("ECGPU_VM3_GEN_ARGS", "--lanes 16 --lanes-c 12 --window 60").split()
# -pragma-unroll-threshold: the sums of products (bls_fp.h fp_sumprod) are 13 rows x up to 13 x 13 multiply-adds that
# must be fully unrolled for their column accumulators to stay in registers; the default threshold stops at ~1000.
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-mllvm", "-pragma-unroll-threshold=1000000", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + (["-DECG_EXPERIMENTS"] if EXPERIMENTS else [])


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    out = [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h")]
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            out.append(os.path.join(CSRC, f))
    return out


_INC = None


def _deps(src: str, seen=None):
    """files `src` includes with quotes, transitively (the .hip second-build units include a .hip; ECG_VM3_PROG_HEADER is a
    macro include): an edit to one header rebuilds only the objects that see it"""
    import re
    global _INC
    if _INC is None:
        _INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)
    seen = seen if seen is not None else set()
    if src in seen or not os.path.exists(src):
        return seen
    seen.add(src)
    text = open(src).read()
    names = _INC.findall(text)
    if "ECG_VM3_PROG_HEADER" in text:
        names.append("bls_vm3_prog.h")
    for n in names:
        for base in (os.path.dirname(src), CSRC, os.path.join(ROOT, "include")):
            cand = os.path.normpath(os.path.join(base, n))
            if os.path.exists(cand):
                _deps(cand, seen)
                break
    return seen


def source_hash(src: str) -> str:
    """sha256 over the contents of `src` and of every file it includes (transitively) plus the compiler flags: the identity of
    the kernels of one translation unit.  Recorded per object in lib/build_manifest.json when the object is built; bench.py
    only quotes a PMC pass (profiles/pmc_traffic.json) whose recorded hash equals the manifest's, i.e. that was taken on the
    very kernels that are loaded."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS[:5] + FLAGS[7:9]).encode())
    for f in sorted(_deps(src)):
        h.update(os.path.relpath(f, ROOT).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(hashlib.sha256(fh.read()).digest())
    return h.hexdigest()[:16]


MANIFEST = os.path.join(LIBDIR, "build_manifest_exp.json" if EXPERIMENTS else "build_manifest.json")


def _write_manifest(entries: dict, path: str = MANIFEST):
    import json
    try:
        with open(path) as f:
            cur = json.load(f)
    except (OSError, ValueError):
        cur = {}
    cur.update(entries)
    with open(path + ".tmp", "w") as f:
        json.dump(cur, f, indent=1, sort_keys=True)
    os.replace(path + ".tmp", path)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build failed: " + cmd[-1])
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


def generate_vm_programs(verbose: bool = True) -> str:
    """csrc/bls_vm3_prog.h (generated tables) is produced by tools/gen_bls_vm3.py, not committed."""
    out = None
    for script, header, gen_args in (("gen_bls_vm3.py", "bls_vm3_prog.h", VM3_GEN_ARGS),):
        gen = os.path.join(ROOT, "tools", script)
        out = os.path.join(CSRC, header)
        stamp = out + ".args"
        want = " ".join(gen_args)
        have = open(stamp).read() if os.path.exists(stamp) else None
        if _newer(out, [gen]) or have != want:
            if verbose:
                print("[ecgpu build] generating csrc/" + header, flush=True)
            r = subprocess.run([sys.executable, gen] + gen_args, capture_output=True, text=True)
            if r.returncode != 0:
                sys.stderr.write(r.stderr)
                raise RuntimeError("tools/" + script + " failed")
            with open(out + ".tmp", "w") as f:
                f.write(r.stdout)
            os.replace(out + ".tmp", out)
            with open(stamp, "w") as f:
                f.write(want)
    return out


def build_lib(verbose: bool = True) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    generate_vm_programs(verbose)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    jobs = []
    objs = []
    built = {}
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJDIR, f[:-4] + ".o")
        objs.append(obj)
        if _newer(obj, sorted(_deps(src)) + [os.path.abspath(__file__)]):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])
            built[f] = source_hash(src)
    if jobs:
        if verbose:
            print(f"[ecgpu build] compiling {len(jobs)} object(s) for {ARCH}", flush=True)
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if not os.path.exists(MANIFEST):  # objects of an older build: they are up to date, so their sources are the present ones
        built = {f: source_hash(os.path.join(CSRC, f)) for f in srcs}
    if built:
        _write_manifest(built)
    if jobs or _newer(LIB, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
        if verbose:
            print("[ecgpu build] linked", LIB, flush=True)
    return LIB


def build_hostsim(verbose: bool = True, variant: str = "") -> str:
    """g++ build of the same csrc headers: CPU-side kernel simulator for tests ONLY.  variant "calls": the compact-code
    build of the G2 stage kernels (-DECG_TOWER_CALLS), a library of its own."""
    d = os.path.join(ROOT, "tests", "hostsim")
    generate_vm_programs(verbose)
    srcs = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".cpp"))
    tag = "_" + variant if variant else ""
    defs = ["-DECG_TOWER_CALLS"] if variant == "calls" else []
    out = os.path.join(d, f"libhostsim{tag}.so")
    os.makedirs(os.path.join(d, "obj"), exist_ok=True)
    objs, jobs = [], []
    for src in srcs:
        obj = os.path.join(d, "obj", os.path.basename(src)[:-4] + tag + ".o")
        objs.append(obj)
        if _newer(obj, [src] + _headers()):
            jobs.append(["g++", "-O2", "-std=c++17", "-fPIC", "-pthread", "-c", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas"] + defs
                        + ["-I" + CSRC, "-I" + os.path.join(ROOT, "include"), src, "-o", obj])
    if jobs:
        if verbose:
            print("[ecgpu build] building tests/hostsim" + tag, flush=True)
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if jobs or _newer(out, objs):
        _run(["g++", "-shared", "-pthread", "-o", out] + objs)
    return out


def build_oracle_c(verbose: bool = True) -> str:
    d = os.path.join(ROOT, "oracle")
    if os.path.exists(os.path.join(d, "Makefile")):
        _run(["make", "-s", "-C", d])
    return d


if __name__ == "__main__":
    build_lib()
    if "--hostsim" in sys.argv:
        build_hostsim()
    if "--oracle" in sys.argv:
        build_oracle_c()
