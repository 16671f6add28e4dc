#!/usr/bin/env python3
"""BASELINE config C5 on ONE GPU's shard (per-GPU batch 1): a bf16 causal-attention transformer block forward + backward
through the Python operator API and Tensor.backward - d_model 4096, 32 heads x 128, S 4096, gated MLP of width 16384, only
operators the reference API has. Reports ms per step, tokens/s and the TFLOP/s of the block's matrix work
(GEMMs 6·T·d·(4d + 3f), attention 14·S²·d / 2 per batch element); per-kernel times from the library's profiling mode."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import kfunca_amd as kfunca  # noqa: E402
from kfunca_amd import hip_abi as H  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--json", default="")
    ap.add_argument("--graph", action="store_true", help="capture one step into a HIP graph and time its replays")
    ap.add_argument("--form", choices=["reference", "fused", "fused-norm"], default="reference",
                    help="reference: only operators the reference API has; fused: attention on the packed QKV projection + GEMM tails "
                         "(same math); fused-norm: that plus the two rms_norms of a real pre-norm block")
    args = ap.parse_args()
    B, S, Hh, D, f = 1, 4096, 32, 128, 16384
    d, T = Hh * D, B * S
    rng = np.random.default_rng(1005)

    def param(shape, scale):
        t = kfunca.from_numpy((rng.uniform(-1, 1, shape) * scale).astype(np.float32), 0).bfloat16()
        t.set_requires_grad(True)
        return t

    x = param((T, d), 1.0)
    w = [param(s, 1.0 / np.sqrt(s[0])) for s in ((d, 3 * d), (d, d), (d, f), (d, f), (f, d))]
    g = kfunca.from_numpy(rng.uniform(-1, 1, (T, d)).astype(np.float32), 0).bfloat16()
    gains = [param((d,), 1.0) for _ in range(2)]

    def step_fused():
        n1 = kfunca.rms_norm(x, gains[0], 1e-5) if args.form == "fused-norm" else x
        a = kfunca.causal_attention_qkv(kfunca.gemm(n1, w[0], 1.0, 0.0), B, S, Hh)
        h = kfunca.gemm_fused(a, w[1], 1.0, None, None, x)
        n2 = kfunca.rms_norm(h, gains[1], 1e-5) if args.form == "fused-norm" else h
        up = kfunca.gemm(n2, w[3], 1.0, 0.0)
        y = kfunca.gemm_fused(kfunca.gemm_fused(n2, w[2], 1.0, None, up, None), w[4], 1.0, None, None, h)
        y.backward(g)

    def step_reference():
        qkv = kfunca.gemm(x, w[0], 1.0, 0.0)
        q, k, v = (t.contiguous().view(B, S, Hh, D).permute(0, 2, 1, 3).contiguous() for t in qkv.split([d, d, d], 1))
        a = kfunca.causal_attention(q, k, v).permute(0, 2, 1, 3).contiguous().view(T, d)
        h = x + kfunca.gemm(a, w[1], 1.0, 0.0)
        y = h + kfunca.gemm(kfunca.gemm(h, w[2], 1.0, 0.0) * kfunca.gemm(h, w[3], 1.0, 0.0), w[4], 1.0, 0.0)
        y.backward(g)

    body = step_reference if args.form == "reference" else step_fused

    def step():  # one training step's shape: gradients start empty (an optimizer's zero_grad), forward, backward
        for t in [x] + w + gains:
            t.zero_grad()
        body()

    for _ in range(args.warmup):
        step()
    kfunca.synchronize(0)
    graph = None
    if args.graph:  # the allocator is warm: the captured step re-uses cached blocks, nothing is hipMalloc'ed while recording
        kfunca.graph_begin(0)
        step()
        graph = kfunca.graph_end(0)
        kfunca.graph_launch(graph, 0)
        kfunca.synchronize(0)
    H.profile_reset()
    H.profile_enable(not args.graph)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if graph is not None:
            kfunca.graph_launch(graph, 0)
        else:
            step()
    kfunca.synchronize(0)
    t1 = time.perf_counter()
    H.profile_enable(False)
    ms = (t1 - t0) / args.steps * 1e3
    flops = 6.0 * T * d * (4 * d + 3 * f) + 14.0 * B * S * S * d / 2
    prof = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps} for k, v in H.profile_results().items()}
    ew = sum(v["launches_per_step"] for k, v in prof.items() if k.startswith("ew_"))
    out = {"mode": "hip graph replay" if args.graph else "eager", "form": args.form, "elementwise_launches_per_step": ew, "config": "C5 shard: bf16 block fwd+bwd, B=1 S=4096 d=4096 H=32 D=128 f=16384, Python operator API + Tensor.backward",
           "ms_per_step": ms, "tokens_per_s": T / (ms * 1e-3), "matrix_tflops": flops / (ms * 1e-3) / 1e12,
           "device_ms_per_step": sum(v["ms_per_step"] for v in prof.values()), "kernels": prof}
    print(json.dumps(out, indent=1))
    if args.json:
        Path(args.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
