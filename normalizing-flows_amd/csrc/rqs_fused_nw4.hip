// rqs_fused_nw4.hip -- the second build of rqs_fused.hip: the inference kernel on 4-wave (128-row) workgroups, under the name
// rqs_fused_kernel_nw4, and its chain dispatch nf_rqs_fused_chain_nw4_ -- what nf_rqs_fused_chain runs for batches of at most 32 768 rows
// (round 6, late; see the note at nf_rqs_fused_chain).  Same source, same arithmetic per row: a row's result does not depend on the
// workgroup size (tests/test_gpu_parity.py::test_small_batch_workgroups_give_the_same_bits).  (Counted waits: NF_WAIT_VMCNT through the
// included file -- the -DNF_SAFE_WAITS lint build recompiles this unit too.)
#define NF_FUSED_WAVES 4
#define NF_FUSED_SECONDARY 1
#include "rqs_fused.hip"
