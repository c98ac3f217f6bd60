/* cova_probe.h -- micro-benchmark probes of tools/ (libcova_probe.so); NOT part of libcova_hip.so. */
#ifndef COVA_PROBE_H
#define COVA_PROBE_H
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ diagnostics (bench tools only)
 * sustained f32-MFMA rate probe: blocks x 8 waves, iters*16 MFMAs (4096 FLOP each) per wave */
int cova_probe_mfma_f32(float *scratch, int blocks, int iters, void *stream);
/* MFMA (16x16x4 f32) chain with valu_per_mfma (0,2,4,6) independent FMAs after each MFMA; mfma=0: VALU only */
int cova_probe_mfma_valu(float *scratch, int blocks, int iters, int valu_per_mfma, int mfma, void *stream);
/* same-wave probe: per wave iters*16 MFMAs interleaved with iters*loads_per_iter (0..2) float4 loads per lane */
int cova_probe_mfma_load(float *scratch, const float *buf, long long n4, int blocks, int iters,
                         int loads_per_iter, void *stream);
/* mixed probe: mfma_blocks MFMA blocks + stream_blocks blocks streaming buf (n4 float4, `passes` times) */
int cova_probe_mix(float *scratch, const float *buf, long long n4, int mfma_blocks, int stream_blocks,
                   int iters, int passes, void *stream);
/* lane-pattern probe: NHWC-64 copy (or load-only) with the Winograd epilogue's lane mapping (mode 0),
 * fully contiguous lanes (1) or 64-byte segments per 4 lanes (2); lds_bytes limits blocks per CU */
int cova_probe_lane_pattern(const float *in, float *out, long long npix, int mode, int loads_only,
                            int blocks, int lds_bytes, void *stream);

/* fills the whole LDS (160 KB) of `blocks` workgroups -- one per CU -- with a bit pattern (tools/poison_check.py POISON_LDS=1) */
int cova_probe_lds_fill(int pattern_bits, int blocks, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* COVA_PROBE_H */
