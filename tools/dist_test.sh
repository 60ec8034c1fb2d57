#!/usr/bin/env sh
# Multi-GPU inference launcher with the reference's calling convention (docs/user_infer.md:113-130):
#     sh tools/dist_test.sh ${CONFIG_FILE} ${GPU_NUM} [arguments of tools/test.py]
# One process per GPU of ONE node over RCCL (torch.distributed backend "nccl"); rendezvous on 127.0.0.1.
# e.g.  sh tools/dist_test.sh configs/patchrefinerv2_zoedepth/v2_mobile_u4k.py 8 --synthetic-weights --cai-mode r32 \
#           --cfg-option general_dataloader.dataset.rgb_image_dir=./examples --save --work-dir ./work_dir/predictions \
#           --test-type general --image-raw-shape 2160 3840 --patch-split-num 4 4 [--shard patches]
set -e
CONFIG=$1
GPUS=$2
shift 2
PORT=${PORT:-29511}
HERE=$(dirname "$0")
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$GPUS" --master-addr 127.0.0.1 --master-port "$PORT" \
    "$HERE/test.py" "$CONFIG" --launcher pytorch "$@"
