// The general backward kernel for plans with lean general groups (hint_plan::has_fly: the d = 100 trees): their rows make the B
// fragments g2' = W3^T g_st themselves, one K <= 4 MFMA per k-block from the thin vectors staged in LDS (hint_rows.hpp row_body
// FLY), instead of a thin phase and a barrier per group.  An instance of its own: the other plans' kernels stay what they were.
#define HINT_BWD_FLY
#define hint_bwd_kernel hint_bwd_kernel_fly
#define launch_bwd launch_bwd_fly
#define set_max_lds_bwd set_max_lds_bwd_fly
#include "hint_bwd.hip"
