// Grouped hi/lo-split bf16 GEMM: up to kGemmGroupMax independent products C = op(A) * op(B) in ONE launch, with the
// split-K combine, the bias / dropout-row epilogues and the bias column sums inside it (gemm_group.hip).  Used by
// comic_decoder_train_step (decoder_exec.hip) for the products that do not feed the recurrence: the memory and rnn-init
// projections before the time loops, every weight gradient after them (src/model_base.py:325-405: tf.gradients of the
// dense layers of common/ops.py:200-238, common/ops_rnn.py:440-447, :545, src/model_base.py:541-543, :618-621).
#pragma once
#include "common.h"

constexpr int kGemmGroupMax = 20;
enum { COMIC_GG_TN = 0, COMIC_GG_NN = 1, COMIC_GG_NT = 2 };

struct ComicGemmProb {
  const float* A;      // TN: [K][M] (lda >= M)   NN, NT: [M][K] (lda >= K)   null with ones_a
  const float* B;      // TN, NN: [K][N] (ldb >= N)   NT: [N][K] (ldb >= K)
  float* C;            // [M][N] (ldc >= N)
  const float* bias;   // [N] added to every row, or null
  const float* mask;   // [M][ld_mask] keep mask: C = (alpha * acc) * mask (DropoutWrapper backward / forward), or null
  int M, N, K, lda, ldb, ldc, ld_mask;
  float alpha, beta;   // C = (alpha * acc + bias) [/ keep * mask] + beta * C
  float keep;          // keep probability of `mask`
  int type;            // COMIC_GG_*
  int ones_a;          // A is all ones: C[0][n] = sum_k B[k][n] (column sums as a product; M must be 1, type TN)
  // set by comic_gemm_group_plan
  int tiles_m, tiles_n, S, k_per_slice, wg_begin;
  int slab_tile0;      // first slab tile (128 x 128 floats) of the problem: partial (tile, slice) is slab tile slab_tile0 + tile * S + slice
  int ticket0;         // first arrival counter of the problem (one per output tile)
};

struct ComicGemmGroup {
  ComicGemmProb p[kGemmGroupMax];
  int n;
  float* slab;         // split-K partial tiles (comic_gemm_group_plan says how many bytes it needs)
  unsigned* tickets;   // one arrival counter per output tile of a split problem; zero before the launch, zero after it
  int xcd_chunk;       // set by comic_gemm_group_launch
};

// Chooses the split of every problem (work items of about equal k length, about `target_items` of them) and lays out
// slab and tickets.  Returns the workgroups of the launch (< 0: error); *slab_bytes / *n_tickets receive what the
// launch needs.
int comic_gemm_group_plan(ComicGemmGroup& g, int target_items, int64_t* slab_bytes, int* n_tickets);
constexpr int kGemmGroupTargetItems = 640;   // work items a grouped launch aims at (split-K degree follows from it)
int comic_gemm_group_launch(const ComicGemmGroup& g, int n_wg, hipStream_t st);
