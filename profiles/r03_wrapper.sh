#!/bin/bash
# the whole wrapper step (VecEnvRLGames.step: clamps, DR noise where configured, task pre + post) at the reference's own sizes
for spec in "Ant 4096" "Humanoid 4096" "Cartpole 512" "ShadowHand 8192" "ShadowHandOpenAI_FF 8192" "AnymalTerrain 2048" "FrankaCabinet 4096" "Quadcopter 8192"; do
  set -- $spec
  for mode in "" "--graph"; do
    timeout -k 10 120 python -m omniisaacgymenvs_amd.scripts.random_policy task=$1 num_envs=$2 --frames 3000 --drop-resets $mode 2>&1 | grep "frames of" | sed "s/^/$1 $2 $mode: /"
  done
done
