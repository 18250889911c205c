// The general backward kernel compiled for rows of at most three tiles (hint_bwd.hip is compiled for four): the plans whose rows
// are no wider - MINIBOONE's: h = 67 is 3 + 2 tiles - run on this one, which spills half as many registers.
#define HINT_NTT 3
#define HINT_PF_DIST 1     // (L2 warm-up ONE consumer ahead in this instance - its plans are MINIBOONE-like: backward 384 -> 377 us, measured once the
                           //  backward warm-up was really wired, round 5; two consumers ahead stays right for the d = 100 plans of the other instances)
#define HINT_NO_ROWDW      // (and without the rows that compute dW1 | db1 themselves: the planner sends those plans to the other instance)
#define hint_bwd_kernel hint_bwd_kernel_n3
#define launch_bwd launch_bwd_n3
#define set_max_lds_bwd set_max_lds_bwd_n3
#include "hint_bwd.hip"
