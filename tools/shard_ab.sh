#!/bin/bash
# A/B of one environment switch on the sharded sort's one-rank rehearsal (GPU box). usage: bash tools/shard_ab.sh "VAR=value" [log2n] [type]
V=${1:-CLO_RADIX_BIG_MIB=128}; L=${2:-28}; T=${3:-uint}
for rep in 1 2; do
  echo "--- default"; python tools/shard_alone_probe.py $L $T 2>&1 | grep "2^"
  echo "--- $V"; env $V python tools/shard_alone_probe.py $L $T 2>&1 | grep "2^"
done
