"""profiles/pmc_traffic.json from the two rocprofv3 --pmc passes of tools/gpu_visit.sh (step `pmc`): per LAUNCH, the mean of
FETCH_SIZE / WRITE_SIZE (KiB) over the dispatches of each dominant kernel.  usage: pmc_to_json.py TAG N_STEPS_IN_PMC_RUN"""
import csv
import glob
import json
import sys

KERNELS = {"k_pairing": "ecg::k_pairing(", "k_miller2": "ecg::k_miller2(", "k_finalexp": "ecg::k_finalexp(", "k_finalexp2": "ecg::k_finalexp2(", "k_vm3_pair_a": "ecg::k_vm3_pair_a(",
           "k_vm3_pair_c": "ecg::k_vm3_pair_c(", "k_merkle_pass<2, ValidatorLeaves>": "k_merkle_pass<2, ecg::ValidatorLeaves>"}


def total(root, counter, needle):
    tot, n = 0.0, 0
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and needle in row["Kernel_Name"]:
                    tot += float(row["Counter_Value"])
                    n += 1
    return tot, n


def src_hash(key):
    """the identity of the kernels the pass was taken on (lib/build_manifest.json, written by the build; bench.py only
    reports a pass whose hash equals the loaded library's)"""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.kernel_source_hash(key)


def main(tag, n_steps, only=None):
    try:
        out = json.load(open("profiles/pmc_traffic.json"))  # passes of other kernel builds stay
    except (OSError, ValueError):
        out = {}
    for key, needle in KERNELS.items():
        if only == "merkle" and "merkle" not in key:
            continue
        f, nf = total(f"gpurun_out/pmc_{tag}_FETCH_SIZE", "FETCH_SIZE", needle)
        w, nw = total(f"gpurun_out/pmc_{tag}_WRITE_SIZE", "WRITE_SIZE", needle)
        if nf and nw:
            # per LAUNCH (mean over the dispatches of the run): bench.py's roofline.traffic is quoted per launch like `achieved`
            out[key] = {"fetch_kib": f / nf, "write_kib": w / nw, "launches_in_run": nf, "src_hash": src_hash(key),
                        "source": f"profiles/{tag}{'_merkle' if only == 'merkle' else ''}_pmc_FETCH_SIZE.txt, profiles/{tag}{'_merkle' if only == 'merkle' else ''}_pmc_WRITE_SIZE.txt (rocprofv3 --pmc, one counter per "
                                  f"pass, bench.py over {n_steps} steps incl. warm-up)"}
    json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else None)
