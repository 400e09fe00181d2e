cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "mlp_block" 2>&1 | tail -30
timeout 900 python -m pytest tests/test_gpu_actor_critic.py -m gpu -q --tb=short -p no:cacheprovider -k "neural_linear_bandit_learn_batch" 2>&1 | tail -30
