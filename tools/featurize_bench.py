"""Measurement of susnet_featurize (SURVEY.md section 8a O2/O3 on caller-supplied rows: state windows, replay batches):
rows of flattened states -> flat / plane features.  Prints rows/s and written GB/s."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("sus-net_amd")
from bench import CONFIGS, POLICY_COMPONENTS, make_env  # noqa: E402


def main():
    spec = CONFIGS["cfg3"]
    env = make_env(pkg, spec, 65536, 3, 0, torch.device("cuda:0"))
    env.reset()
    traj = env.rollout(32, obs=pkg.ObsConfig("raw", dtype=torch.uint8))
    rows_u8 = traj["obs"].reshape(-1, env.flattened_state_size).contiguous()  # 2.1 M rows of real states
    for name, oc, rows in (("flat f32 (policy components) from u8 rows", pkg.ObsConfig("flat", POLICY_COMPONENTS, dtype=torch.float32), rows_u8),
                           ("flat f32 from f32 rows (replay batch)", pkg.ObsConfig("flat", POLICY_COMPONENTS, dtype=torch.float32), rows_u8.float()),
                           ("planes f32 from u8 rows", pkg.ObsConfig("planes", dtype=torch.float32), rows_u8[: 1 << 19])):
        for _ in range(3):  # (the caching allocator needs two live output blocks before a call stops reaching hipMalloc)
            out = env.featurize(rows, oc)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            out = env.featurize(rows, oc)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 5
        o1 = out[0] if isinstance(out, (tuple, list)) else out
        nbytes = sum(x.numel() * x.element_size() for x in (out if isinstance(out, (tuple, list)) else [out]) if x is not None)
        print(json.dumps({"what": name, "rows": int(rows.shape[0]), "features_per_row": int(o1[0].numel()), "ms": dt * 1e3,
                          "rows_per_s": rows.shape[0] / dt, "written_GBs": nbytes / dt / 1e9,
                          "read_GBs": rows.numel() * rows.element_size() / dt / 1e9}))


if __name__ == "__main__":
    main()
