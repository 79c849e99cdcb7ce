"""frame time of a Monte-Carlo stream bank (bench.py's stream_bank_numbers on chosen cases)
    python tests/tools/time_bank.py uarm_S1024_mc50_T6 [watch_S1024_mc25_T8 ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

ALL = {"S1024_mc1": ("S1024_mc1", "pocket", 1024, None, 1, 100), "S1024_mc25": ("S1024_mc25", "pocket", 1024, 25, 1, 30),
       "S8192_mc25": ("S8192_mc25", "pocket", 8192, 25, 1, 8), "uarm_S1024_mc50_T6": ("uarm_S1024_mc50_T6", "uarm", 1024, 50, 1, 12),
       "uarm_S4096_mc50_T6": ("uarm_S4096_mc50_T6", "uarm", 4096, 50, 1, 6),
       "watch_S1024_mc25_T8": ("watch_S1024_mc25_T8", "watch", 1024, 25, 10, 20)}
if __name__ == "__main__":
    import torch
    from wear_mocap_ape_amd.utility.names import NNS_INPUTS, NNS_TARGETS
    torch.cuda.set_device(0)
    cases = [ALL[k] for k in (sys.argv[1:] or ["uarm_S1024_mc50_T6"])]
    model = bench._bank_model(bench.POCKET, (NNS_INPUTS.WATCH_PHONE_CAL_HIP, NNS_TARGETS.ORI_CAL_LARM_UARM_HIPS))
    out = bench.stream_bank_numbers(model, None, cases=cases)
    if os.environ.get("BRIEF"):
        for k, v in out.items():
            if isinstance(v, dict):
                r = v["roofline"]
                print(f"{k}: frame {v['ms_per_frame_of_all_streams']:.4f} ms, kernel {r['kernel']} {r['kernel_ms']:.4f} ms, "
                      f"frac {r['frac']:.3f} executed {r.get('frac_executed', 0):.3f}")
            else:
                print(k, v)
    else:
        print(json.dumps(out, indent=1))
