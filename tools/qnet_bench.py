"""Time the policy network of BASELINE config 5 on 65 536 cfg3 environments: the fused kernel (susnet_qnet_forward) against the torch
module (hipBLASLt GEMMs + PReLU kernels) on the same observation.  One JSON line."""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("sus-net_amd")


def timed(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n  # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    comps = ["onehot_pos", "alive_crew", "closest_crew"]
    import bench
    env = bench.make_env(pkg, bench.CONFIGS["cfg3"], B, 1, 0, torch.device("cuda:0"), pkg.ObsConfig("flat", comps))
    env.reset()
    model = pkg.policy.reference_imposter_mlp(env, comps, seed=0)
    net = pkg.policy.pack_mlp(env, model, comps)
    spatial = torch.zeros(B, 1, 1, device=env.device)
    with torch.no_grad():
        t_torch = timed(lambda: model(spatial, env.obs))
        t_fused = timed(lambda: env.qnet_forward(net))
        err = float((env.qnet_forward(net) - model(spatial, env.obs)).abs().max())
    flops = 2.0 * B * sum(a * b for a, b in zip(net.dims[:-1], net.dims[1:]))
    mfma_flops = 2.0 * B * (256 * 128 + 128 * 64 + 64 * 32 + 32 * 32)
    print(json.dumps({"envs": B, "dims": net.dims, "torch_us": round(t_torch, 1), "fused_us": round(t_fused, 1),
                      "model_tflops_fused": round(flops / t_fused / 1e6, 1), "mfma_tflops_issued": round(mfma_flops / t_fused / 1e6, 1),
                      "mfma_peak_f32_tflops": 157.3, "max_abs_diff": err}))


if __name__ == "__main__":
    main()
