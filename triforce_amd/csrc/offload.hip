// Offloading tier: pinned host <-> device KV page streaming for the TP engine.
// Replaces DistributedKVCacheBuffer.copy_kv (models/cache.py:372-376: whole-layer H2D `copy_(non_blocking)`),
// DistributedSimpleCache.copy_back_from_buffer (:345-351: D2H of the new tokens) and the host part of
// DistributedRetrievalCache.update_graph_cache (:573-575).  The reference brackets every offloaded layer
// with two device-wide torch.cuda.synchronize() calls (TP_llama.py:222,228); here the copies are
// hipMemcpy2DAsync on a dedicated copy stream and ordering is expressed with events by the caller.
//
// A "KV block" is H rows (heads) of `width_elems` contiguous fp16 (n tokens x D) at a pitch of
// `pitch_elems` (T x D) — the head-major layout; only the live [0, seq_len) tokens of a layer travel.
#include "common.h"

extern "C" int tf_kv_h2d_async(void* dst_dev, int64_t dst_pitch_elems, const void* src_host,
                               int64_t src_pitch_elems, int64_t width_elems, int H, void* copy_stream) {
    if (!dst_dev || !src_host || width_elems < 0 || H < 1) return TF_EINVAL;
    if (width_elems == 0) return TF_OK;
    hipError_t e = hipMemcpy2DAsync(dst_dev, (size_t)dst_pitch_elems * 2, src_host, (size_t)src_pitch_elems * 2,
                                    (size_t)width_elems * 2, (size_t)H, hipMemcpyHostToDevice,
                                    (hipStream_t)copy_stream);
    return e == hipSuccess ? TF_OK : (int)e;
}

extern "C" int tf_kv_d2h_async(void* dst_host, int64_t dst_pitch_elems, const void* src_dev,
                               int64_t src_pitch_elems, int64_t width_elems, int H, void* copy_stream) {
    if (!dst_host || !src_dev || width_elems < 0 || H < 1) return TF_EINVAL;
    if (width_elems == 0) return TF_OK;
    hipError_t e = hipMemcpy2DAsync(dst_host, (size_t)dst_pitch_elems * 2, src_dev, (size_t)src_pitch_elems * 2,
                                    (size_t)width_elems * 2, (size_t)H, hipMemcpyDeviceToHost,
                                    (hipStream_t)copy_stream);
    return e == hipSuccess ? TF_OK : (int)e;
}
