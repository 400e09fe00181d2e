/* Prints sizeof() of every struct of include/pearl_amd.h so the ctypes mirror can be checked. */
#include <stddef.h>
#include <stdio.h>

#include "../include/pearl_amd.h"

#define P(T) printf(#T " %zu\n", sizeof(T))
#define O(T, m) printf(#T "." #m " %zu\n", offsetof(T, m))

int main(void) {
  P(pa_arena_desc); P(pa_transition); P(pa_columns); P(pa_batch_out); P(pa_dqn_desc);
  P(pa_dqn_buffers); P(pa_dqn_batch); P(pa_learn_args);
  O(pa_arena_desc, staging_rows); O(pa_transition, terminated); O(pa_columns, avail_bcast);
  O(pa_batch_out, rep_dim); O(pa_batch_out, rep_onehot); O(pa_dqn_desc, lr); O(pa_dqn_desc, amsgrad);
  O(pa_dqn_batch, x); O(pa_dqn_batch, next_avail_bcast); O(pa_dqn_batch, next_action_rep);
  O(pa_dqn_desc, double_q); O(pa_learn_args, training_steps0);
  O(pa_learn_args, seed); O(pa_learn_args, losses_out); O(pa_learn_args, idx_host);
  return 0;
}
