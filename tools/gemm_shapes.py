#!/usr/bin/env python3
"""Shape study: histogram of the (kernel, M, N, K) launches of a SVLA_GEMM_LOG (+ SVLA_GEMM_LOG_ALL=1) file.
  SVLA_GEMM_LOG=/tmp/g.log SVLA_GEMM_LOG_ALL=1 python tools/acting_probe.py; python tools/gemm_shapes.py /tmp/g.log"""
import collections, sys
c = collections.Counter()
for line in open(sys.argv[1]):
    k, M, N, K, *x = line.split()
    c[(k, int(M), int(N), int(K), int(x[0]) if x else 0)] += 1
tot = sum(2.0 * M * N * K * n for (k, M, N, K, x), n in c.items())
print(f"# {sum(c.values())} launches, {tot/1e12:.2f} TFLOP")
for (k, M, N, K, x), n in sorted(c.items(), key=lambda kv: -2.0 * kv[0][1] * kv[0][2] * kv[0][3] * kv[1]):
    print(f"{k:28s} M={M:8d} N={N:5d} K={K:5d} extra={x:5d} n={n:6d}  {2.0*M*N*K*n/1e9:10.1f} GFLOP  {100*2.0*M*N*K*n/tot:5.1f} %")
