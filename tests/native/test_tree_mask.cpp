// Host check of tf_tree_vis8 (triforce_amd/csrc/tree_mask.h) against the per-key rule, exhaustively over offsets.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../triforce_amd/csrc/tree_mask.h"

static bool naive(const uint32_t* row, int kidx, int sk, int tree_start) {
    if (kidx >= sk) return false;
    const int j = kidx - tree_start;
    if (j < 0) return true;
    return (row[j >> 5] >> (j & 31)) & 1u;
}

int main() {
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed; };
    long checked = 0;
    for (int words = 1; words <= 17; words += 4) {
        std::vector<uint32_t> row(words);
        for (int rep = 0; rep < 6; ++rep) {
            for (auto& w : row) w = (rep == 0) ? 0u : (rep == 1) ? 0xFFFFFFFFu : (rnd() ^ (rnd() << 7));
            for (int tree_start = 0; tree_start <= 70; ++tree_start)
                for (int tree = 1; tree <= 32 * words; tree += (tree < 40 ? 1 : 37)) {
                    const int sk = tree_start + tree;
                    for (int kidx0 = 0; kidx0 < sk + 16; ++kidx0) {           // every alignment, including past the end
                        const uint32_t got = tf_tree_vis8(row.data(), words, kidx0 - tree_start, sk - kidx0);
                        for (int r = 0; r < 8; ++r) {
                            const bool want = naive(row.data(), kidx0 + r, sk, tree_start);
                            if (((got >> r) & 1u) != (want ? 1u : 0u)) {
                                std::printf("MISMATCH words=%d rep=%d tree_start=%d sk=%d kidx0=%d r=%d got=%02x\n", words,
                                            rep, tree_start, sk, kidx0, r, got);
                                return 1;
                            }
                            ++checked;
                        }
                        if (got >> 8) { std::printf("HIGH BITS SET\n"); return 1; }
                    }
                }
        }
    }
    std::printf("OK %ld key checks\n", checked);
    return 0;
}
