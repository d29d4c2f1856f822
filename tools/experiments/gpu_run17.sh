#!/bin/bash
# Record of the E-step bottleneck experiments (round 1, config 3, one box per table).  The builds were
# made with temporary macros in k_e_step that are no longer in the tree:
#   PLSA_EXP_FIXW=4096   word ids masked to 4096 rows (1 MB of P(w|z): every gather an L2 hit)
#   PLSA_EXP_FIXDW=1024  both tables masked to 1024 rows (256 KB each)
#   PLSA_EXP_NOSTORE     P stores predicated off (inv < 0 never holds)
#   PLSA_EXP_NOLOAD      gathers replaced by arithmetic on the ids
#   PLSA_EXP_NOU         only the P(z|d) gather replaced
#   hipcc ... -DPLSA_EXP_...=1 enstop_amd/csrc/plsa_hip.hip -o build/exp/lib_X.so ; ENSTOP_AMD_LIB=... python bench.py --steps 10 --warmup 2
#
#   flat kernel as shipped            6.85 ms   (this box; 6.27-6.51 on others)
#   FIXW                              6.63 ms   -> misses of the topic table are not the problem
#   NOSTORE                           1.72 ms   -> read side alone
#   NOSTORE + FIXW                    1.71 ms
#   NOLOAD                            4.09-4.25 ms -> store stream alone = non-temporal fill rate
#   FIXDW (all gathers hit L2)        5.89-5.93 ms -> loads cost even when they hit
#   NOU (no P(z|d) gathers)           5.71 ms   against 6.29-6.30 ms on that box
#   software pipeline (next gathers before current stores), UNR 2/4/8: 6.46-6.69 ms against 6.55-6.74 ms (noise)
#   document-owned kernel, gathers per burst 2 / 4 / 8 / 16:  5.57-5.61 / 5.81-5.82 / 5.86-5.87 / 5.42-5.43 ms
#   document-owned + software pipeline (burst 16): 5.55 ms;  launch_bounds hint of 2 waves: 5.43 ms
#   config 5 (k = 128, 2 chunks per lane) burst 4 / 8 / 16 float4 per lane: 50.4 / 49.4 / 48.9 ms  (flat kernel: 58.7 ms)
#   config 2 / 1 document-owned: 0.363 / 0.145 ms against flat 0.33-0.34 / 0.084 ms -> size rule in run_e_step
#   fill probe in the E-step's store order (each wave 16 consecutive 1-KB rows): 5.95-6.35 TB/s against 6.4-6.6 TB/s grid-stride
echo "record only"
