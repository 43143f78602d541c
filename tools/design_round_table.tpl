**Measured (round 6, 1×MI355X, `profiles/r6_*`; the box of the collection - the pool's boxes differ by ±2-4 %: THIS box's vocoder-alone step is @VOCA@ ms; round 5's collection box: 1.199 with round 5's library; `profiles/r6_voc_kernels.txt` has the round's kernels on a fast box).**

| | bf16-limb (default, headline) | exact-f32 MFMA (`--arith f32`) |
|---|---|---|
| B = 64 streams, full pipeline, pipelined steps | **@MS64@ ms per step = @V64@ k chunks/s** (@RT64@ real-time streams; unprimed @UNP64@; the driver's K = 20 / W = 5: @K20@; 1.33-1.40 across the boxes of this round's runs, whose vocoder-alone times range from 1.19 to 1.25 ms) | @F32MS@ ms |
| same, blocking fused steps | **p50 @P50@ ms** (p95 @P95@; round 5: 1.81 on its box) | p50 @F32P50@ ms |
| B = 1 / B = 4 streams, blocking p50 | **@B1@ / @B4@ ms** per 80 ms chunk (round 5: 0.715 / 0.793); pipelined steps of 1 / 2 / 4 streams **0.40 / 0.43 / 0.55 ms** (round 5: 0.635 / 0.667 / 0.695; `profiles/r6_stream_sweep.txt`) | @F32B1@ |
| dominant kernel (9 launches per step, 105.7 GFLOP algorithmic) | `cnk::resblock_limb_kernel`: @KMS@ ms = **@ACH@ TFLOP/s = @FRAC@ of the limb ceiling** (419.5); C = 128 / 64 / 32: @F128@ / @F64@ / @F32C@ | `cnk::resblock_fused_kernel` @F32FRAC@ of 157.3 |
| stages alone (`profiles/r6_stage_times.txt`) | Emformer @EMF@, decoder @DEC@, vocoder @VOC@ ms; pipelined @PIPE@ | — |
| front-end cost (`frontend_cost_ms`) | @FE@ ms (round 5: 0.144) | — |
| whole step against both rooflines | 168.3 GFLOP / step = @STF@ TFLOP/s (@SMF@ of the f32 MFMA peak); 245 MB algorithmic = 0.02 of HBM peak | — |
| HBM bytes per step (PMC, gfx950-corrected) | **@HBM@ GB** (fetch @FETCH@ + write @WRITE@): decoder launch @DECF@ MB (group-fastest layout, §4.2; 159 MB member-fastest: measured, not shipped - §4.6), Emformer 78 MB | — |
| other BASELINE configs | b1win @B1WIN@ ms per blocking windowed step; b128s2 (40 ms chunks, B = 128) **@B128MS@ ms per step = @B128V@ k chunks/s**, p50 @B128P50@; b128s2win @B128WIN@ ms; b128s2mem4 @MEM4@ ms | — |
| gather choreography on one rank (`CONAN_BENCH_COMM=1`) | @COMM@ ms per step | — |
| `CONAN_STREAMS_FIXED_PLAN` at 64 of 64 slots active (`profiles/r6_fixed_plan_cost.txt`, three alternating runs, medians) | @FPD@ ms per step default against @FPF@ fixed (ups.1 without its split tail, on the f32 MFMA; vocoder alone @FPVD@ against @FPVF@; blocking p50 @FPPD@ against @FPPF@) | — |
| CPU baseline (oracle, AMD EPYC 9575F host, 16 threads chosen by probe) | reference-semantics loop @CPU@ chunks/s, stateful @CPUS@ | — |
| GPU tests | @TESTS@ | — |

