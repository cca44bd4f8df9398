#!/bin/bash
# after the task-batching change of host/pipeline.py: the frame-shard, pipeline and golden model cases again
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c41; mkdir -p $O
export TMPDIR=/tmp
( timeout 200 python -m pytest tests/test_model_gpu.py tests/test_frame_shard.py tests/test_reference_protocol_gpu.py -m gpu -q -x -k "shard or pipeline_spatial or pipeline_temporal or golden or protocol or prune" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
