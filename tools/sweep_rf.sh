#!/bin/bash
# usage: tools/sweep_rf.sh "56:48 40:32 24:16 0:0"
cd "$(dirname "$0")/.."
for cfg in $1; do
  rf=${cfg%%:*}; rfb=${cfg##*:}
  (cd automatic-speech-recognition_amd/csrc && rm -f build/rnn_seq.o && make CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLAS_RF_LSTM=$rf -DLAS_RFB_LSTM=$rfb" >/dev/null 2>&1)
  echo "== RF=$rf RFB=$rfb"
  python tools/bench_rnn.py 2>&1 | grep cell=
done
