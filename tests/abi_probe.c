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
  P(pa_mlp_desc); P(pa_mlp_buffers); P(pa_sac_step_args); P(pa_ddpg_step_args); P(pa_ac_loop_args);
  O(pa_mlp_desc, max_batch); O(pa_mlp_desc, lr); O(pa_mlp_desc, identity_layers); O(pa_mlp_desc, hidden_act); O(pa_mlp_desc, layer_norm); O(pa_mlp_desc, batch_norm); O(pa_mlp_desc, residual);
  O(pa_mlp_buffers, max_exp_avg_sq);
  O(pa_sac_step_args, ld_state); O(pa_sac_step_args, terminated); O(pa_sac_step_args, noise_critic);
  O(pa_sac_step_args, target_entropy); O(pa_sac_step_args, alpha_lr); O(pa_sac_step_args, alpha_amsgrad);
  O(pa_sac_step_args, alpha_step); O(pa_sac_step_args, B); O(pa_sac_step_args, gamma);
  O(pa_sac_step_args, actor_step); O(pa_sac_step_args, scratch); O(pa_sac_step_args, log_prob_out);
  O(pa_ddpg_step_args, target_noise); O(pa_ddpg_step_args, noise_clip); O(pa_ddpg_step_args, low);
  O(pa_ddpg_step_args, zeros); O(pa_ddpg_step_args, B); O(pa_ddpg_step_args, gamma);
  O(pa_ddpg_step_args, do_actor); O(pa_ddpg_step_args, critic_tau); O(pa_ddpg_step_args, actor_step);
  O(pa_ddpg_step_args, losses);
  O(pa_ac_loop_args, idx_lists); O(pa_ac_loop_args, batch); O(pa_ac_loop_args, noise);
  O(pa_ac_loop_args, noise_stride); O(pa_ac_loop_args, losses); O(pa_ac_loop_args, losses_stride);
  O(pa_ac_loop_args, actor_update_freq); O(pa_ac_loop_args, training_step0);
  O(pa_ac_loop_args, gather_rounds);
  O(pa_batch_out, curr_avail_rep);
  P(pa_bandit_step_args); P(pa_ppo_learn_args); P(pa_dsac_step_args); P(pa_iql_step_args);
  O(pa_bandit_step_args, adam_step); O(pa_bandit_step_args, d); O(pa_bandit_step_args, b_snap);
  O(pa_bandit_step_args, side_stream); O(pa_bandit_step_args, l2_reg_lambda);
  O(pa_bandit_step_args, singular);
  O(pa_ppo_learn_args, idx_lists); O(pa_ppo_learn_args, plane_stride); O(pa_ppo_learn_args, epsilon);
  O(pa_ppo_learn_args, losses_stride); O(pa_ppo_learn_args, critic_step);
  O(pa_dsac_step_args, xq); O(pa_dsac_step_args, curr_rep_bstride); O(pa_dsac_step_args, next_mask);
  O(pa_dsac_step_args, tau); O(pa_dsac_step_args, target_entropy); O(pa_dsac_step_args, alpha_lr);
  O(pa_dsac_step_args, critic_step); O(pa_dsac_step_args, h_out);
  O(pa_iql_step_args, actor_kind); O(pa_iql_step_args, ld_xq); O(pa_iql_step_args, high);
  O(pa_iql_step_args, pick_actor); O(pa_iql_step_args, tau); O(pa_iql_step_args, critic_step);
  O(pa_iql_step_args, losses);
  return 0;
}
