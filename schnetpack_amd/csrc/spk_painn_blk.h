// Block kernels of the PaiNN message for large lists (spk_painn_blk.hip): what the dispatcher in spk_painn.hip calls.
#pragma once
#include "spk_painn_msg.h"

#ifndef SPK_TRY
#define SPK_TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)
#endif

// true if the block kernels cover this shape / list (a usable plan hangs on the graph, F % 16 == 0, n_rbf <= 32, ...)
bool spk_painn_blk_ok(const MsgArgs& a, bool bwd);
// per-call edge tables (A, A', records) from r_ij: once per force call (the drivers) or per message call (a.blocks_prepared == 0)
int spk_painn_blk_prep(const MsgArgs& a, hipStream_t stream);
int spk_painn_blk_fwd(const MsgArgs& a, hipStream_t stream);
int spk_painn_blk_bwd(const MsgArgs& a, hipStream_t stream);
