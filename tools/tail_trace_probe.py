"""Development probe (variant build: tools/build_variant.sh tailtrace "-DECG_TAIL_TRACE" merkle.hip; run with ECGPU_LIB set to it):
where the fused tail of a BeaconState root (k_state_tail) spends its time -- 100 MHz timestamps of the critical field's tile
stage, its finishing job, the last arrival at the state container and the root, plus the completion time of every unit."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ethereum_consensus_amd import _lib, synthetic  # noqa: E402


def main(n=1 << 20):
    L = _lib.load()
    f = synthetic.state_fields(n, "mainnet", seed=5)
    enc = synthetic.serialize_state(f)
    d = torch.frombuffer(bytearray(enc), dtype=torch.uint8).cuda()
    d_root = torch.empty(32, dtype=torch.uint8, device="cuda")
    fs = int(L.ecgpu_beacon_state_deneb_fixed_size(0))
    hfix = (ctypes.c_uint8 * fs).from_buffer_copy(bytes(enc[:fs]))
    s = torch.cuda.current_stream().cuda_stream
    out = (ctypes.c_uint64 * 128)()
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.ecgpu_htr_beacon_state_deneb_dev(ctypes.c_void_p(d.data_ptr()), ctypes.c_uint64(len(enc)), hfix, 0, ctypes.c_void_p(d_root.data_ptr()),
                                                ctypes.c_void_p(s))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        assert rc == 0, rc
        L.ecgpu_debug_tail_trace(out)
        v = list(out)
        t_start = v[0]
        us = lambda x: (x - t_start) / 100.0 if x != 0xFFFFFFFFFFFFFFFF else float("nan")
        print(f"root {it}: host {1e6 * (t1 - t0):.0f} us; tail (us from its first workgroup): critical field's first tile {us(v[1]):.1f}, "
              f"last tile arrived {us(v[2]):.1f}, finishing job done {us(v[3]):.1f}, last arrival at the state container {us(v[4]):.1f}, "
              f"root written {us(v[5]):.1f}")
        if v[3] > v[2]:
            print(f"    finishing job of the critical field: {(v[7] - v[6]) / ((v[3] - v[2]) / 100.0):.0f} shader cycles per us "
                  f"({(v[7] - v[6])} cycles)")
        if it == 5:
            print("  finishing job of field f done at:", " ".join(f"{us(x):.0f}" for x in v[8:8 + 24] if x != 0xFFFFFFFFFFFFFFFF))
            print("  unit j (no tile stage) done at:", " ".join(f"{us(x):.0f}" for x in v[40:40 + 64] if x != 0xFFFFFFFFFFFFFFFF))


if __name__ == "__main__":
    main()
