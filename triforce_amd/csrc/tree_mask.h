// Visibility of 8 CONSECUTIVE keys for one query row of a tree pass, as one byte (bit r = key r visible).
// Plain C++ (no HIP constructs) so that tests/native/test_tree_mask.cpp can check it on the host with g++.
//
// Rule (reference SpecTree_TP.py:65-67,170: dense additive mask over [prefix | tree]): key kidx is visible iff
// kidx < sk and (kidx < tree_start  or  bit (kidx - tree_start) of the row's packed mask is set).
// The per-key form costs a 64-bit address and a 4-byte load per key; 8 consecutive keys are 8 consecutive bits of
// the row, i.e. at most two words — one 64-bit funnel shift.
#pragma once
#include <stdint.h>

#ifndef TF_HD
#ifdef __HIPCC__
#define TF_HD __host__ __device__ __forceinline__
#else
#define TF_HD inline
#endif
#endif

// row: the query row's mask words (words of them); j0 = kidx0 - tree_start (may be negative: prefix keys);
// nvalid = sk - kidx0 (keys r >= nvalid are past the end).
TF_HD uint32_t tf_tree_vis8(const uint32_t* row, int words, int j0, int nvalid) {
    uint32_t vis;
    if (j0 <= -8) {
        vis = 0xFFu;                                        // all eight are prefix keys
    } else {
        const int jj = j0 < 0 ? 0 : j0;                     // first tree column among the eight
        const int w = jj >> 5;
        const uint32_t lo = (w < words) ? row[w] : 0u;
        const uint32_t hi = (w + 1 < words) ? row[w + 1] : 0u;
        const uint64_t two = ((uint64_t)hi << 32) | lo;
        const uint32_t t = (uint32_t)(two >> (jj & 31)) & 0xFFu;          // tree columns jj .. jj+7
        vis = (j0 < 0) ? (((t << (-j0)) | ((1u << (-j0)) - 1u)) & 0xFFu) : t;
    }
    if (nvalid < 8) vis &= (nvalid <= 0) ? 0u : ((1u << nvalid) - 1u);
    return vis;
}
