#!/bin/bash
# gpurun with a small snapshot: the actor-critic / bandit fixtures (80 MB of tests/golden) stay home.
# For DQN-only tests, benches and tools.  usage: tools/gpurun_light.sh <timeout> '<command>'
cd "$(dirname "$0")/.."
cp .gpurunignore /tmp/.gpurunignore.full
{ cat /tmp/.gpurunignore.full; for p in dsac td3 iql ddpg sac ppo squarecb her bootstrap bandit_cfg5 bandit_mae_cfg5 bandit_bce_cfg5; do echo "tests/golden/${p}_*"; done; echo "pearl_amd/csrc/*.o"; } > .gpurunignore
/usr/local/graft/bin/gpurun --timeout ${1:-600} -- "$2"
rc=$?
cp /tmp/.gpurunignore.full .gpurunignore
exit $rc
