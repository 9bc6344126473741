#!/bin/bash
# BASELINE.json configs beyond the headline, one JSON line each (GPU box).  usage: tools/run_configs.sh > out.jsonl
cd "$(dirname "$0")/.."
B="python bench.py --steps 2 --warmup 1 --cpu-shots 300"
$B --code hgp225 --shots 16384 --max-iter 30 2>/dev/null                 # configs[0] shape on the GPU (single window)
$B --code hgp225 --shots 16384 --max-iter 30 --window 3 1 2>/dev/null
$B --code bb72 --shots 65536 2>/dev/null                                   # configs[1]: [[72,12,6]], single window
$B --code bb72 --shots 65536 --window 3 1 2>/dev/null                      # configs[1], W=3 F=1 (6 windows)
$B --window 3 1 --shots 65536 2>/dev/null                                  # headline code, W=3 F=1 (12 windows)
$B --window 5 3 --shots 65536 2>/dev/null                                  # headline code, W=5 F=3 (4 windows)
for p in 0.001 0.002 0.004 0.005 0.006; do $B --p $p --shots 65536 --no-cpu 2>/dev/null; done   # configs[3] p-sweep (1 GPU)
# configs[4]: QLP [[1020,136]], cardinal circuit, R=20, W=3 F=1 (20 windows of 1350 x 18900); LER saturates at p=3e-3, so lower rates too
$B --code qlp1020 --window 3 1 --shots 8192 --no-cpu 2>/dev/null
$B --code qlp1020 --window 3 1 --shots 8192 --p-override 0.001 --cpu-shots 24 2>/dev/null                                   # (CPU leg: 16 x 24 shots)
$B --code qlp1020 --window 3 1 --shots 8192 --p-override 0.0005 --no-cpu 2>/dev/null
$B --code qlp1020 --window 3 1 --shots 4096 --steps 1 --p-override 0.001 --osd-method osd_cs --osd-order 1 --cpu-shots 2 2>/dev/null     # configs[4]: OSD-CS leg (CPU leg: 16 x 2 shots)
$B --code qlp1020 --window 3 1 --shots 4096 --steps 1 --p-override 0.001 --osd-method lsd_0 --no-cpu 2>/dev/null                       # configs[4] code with BP-LSD
$B --osd-method osd_cs --osd-order 1 --shots 32768 --cpu-shots 100 2>/dev/null                                            # headline code with OSD-CS(1)
$B --osd-method lsd_0 --cpu-shots 300 2>/dev/null                                                                         # headline code with BP-LSD (LSD-0)
$B --osd-method lsd_cs --osd-order 1 --cpu-shots 300 2>/dev/null                                                          # ... and with lsd_order 1, what the reference's own BP-LSD calls pass
# the general (one message per edge) BP kernel at the headline code: the reference wrapper's other bp_method / schedule options
$B --bp-method product_sum --schedule serial --max-iter 10 --cpu-shots 100 2>/dev/null
$B --bp-method product_sum --schedule parallel --cpu-shots 100 2>/dev/null
$B --bp-method minimum_sum --schedule serial --max-iter 10 --cpu-shots 100 2>/dev/null
$B --bp-method product_sum --schedule serial --max-iter 10 --osd-method osd_cs --osd-order 1 --shots 16384 --cpu-shots 30 2>/dev/null   # the docs' decoder settings
$B --bp-method product_sum --schedule serial --max-iter 10 --osd-method osd_cs --osd-order 1 --window 5 3 --cpu-shots 100 2>/dev/null  # ... on the docs' windows (W=5 F=3)
$B --bp-method product_sum --schedule serial --max-iter 10 --osd-method osd_cs --osd-order 1 --window 3 1 --cpu-shots 100 2>/dev/null
$B --bp-method product_sum --schedule serial --max-iter 10 --osd-method lsd_cs --osd-order 1 --window 5 3 --no-cpu 2>/dev/null
