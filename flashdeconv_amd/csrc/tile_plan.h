// Static schedule of the tile kernel (tile_kernels.cpp): which lane gathers which gene of a staged 16-spot tile.
//
// The CountSketch sends every gene to one bucket (flashdeconv/core/sketching.py:58-74).  The tile kernel computes the
// bucket sums of 16 spots at a time without atomics: the d buckets are dealt to NW waves x JW groups x 4 lanes-classes
// ("slots"); lane (r, q) of wave w owns, for spot r of the tile, the buckets of slots (w, j, q), j < JW, and walks the
// genes of those buckets in ascending order - exactly the operand layout v_mfma_f64_16x16x4_f64 wants for the
// contraction with X_sketch (B[k = q][n = r]).  The four lanes-classes of a group advance in lockstep, so a group costs
// max_q(count) steps: buckets are grouped by their gene counts (per column block) to keep that padding small.
#pragma once
#include <vector>

namespace fdx {

struct TilePlanHost {
    int G = 0, d = 0, NW = 0, JW = 0, GB = 0, NBLK = 0;
    int NE = 0;                              // entries in the flat tables (4 per step, plus two padding steps per wave)
    int steps = 0;                           // sum of group lengths over all waves, blocks and groups (cost figure)
    int max_wave_steps = 0;                  // steps of the most loaded wave
    int jw_used = 0;                         // groups j >= jw_used hold no bucket in any wave
    std::vector<int> slot_bucket;            // (NW, JW, 4): bucket of slot (w, j, q), -1 = none
    std::vector<unsigned char> len;          // (NW, NBLK, JW): steps of group (w, j) in column block c
    std::vector<int> ent_base;               // (NW, NBLK + 1): first entry of block c of wave w
    std::vector<double> w;                   // (NE): Omega weight of the gene, 0 for padding entries
    std::vector<unsigned short> off;         // (NE): gene index within its column block (a valid one for padding entries)
    std::vector<int> gene;                   // (NE): the gene of the entry, -1 for padding entries
};

// Flat form of the same schedule (tile_kernels.cpp, FF): the steps of a (wave, block) in the dynamic form's order - group by
// group, genes ascending - as one stream: off[(w, c, q)][k] the offset lane class q reads in step k (pad_off past the last step
// and for a lockstep padding slot), gid[(w, c)][k] the group step k adds to.
struct TileFlatHost {
    int NSP = 0;                                 // steps per (wave, block), rounded up to 8 (rows of `off`)
    int GROW = 0;                                // bytes per (wave, block) row of `gid`: 8 (the step count, little endian) + NSP
    std::vector<unsigned short> off;             // (NW, NBLK, 4, NSP)
    std::vector<unsigned char> gid;              // (NW, NBLK, GROW): [0..1] steps of the groups below 16 (a multiple of 8: they come first), [2..3] all steps (likewise), [8 + k] group of step k
};
// `plan` from build_tile_plan; pad_off: the offset a padding slot reads (the kernel keeps weight 0.0 there).
bool build_tile_flat(const TilePlanHost& plan, int pad_off, TileFlatHost* out);

// gene_bucket[g] in [0, d) or -1 (gene not in Omega), gene_w[g] its weight.  GB = genes per column block (the last block
// may be shorter).  Returns false when the shape cannot be scheduled (more than 4*NW*JW buckets, a group longer than 255).
bool build_tile_plan(const int* gene_bucket, const double* gene_w, int G, int d, int NW, int JW, int GB, TilePlanHost* out);


}  // namespace fdx
