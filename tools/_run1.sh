cd /root/repo
python tools/bench_speller.py 2>&1 | grep speller
for m in 1 2 3; do echo abl $m; LAS_LIB_PATH=/root/repo/automatic-speech-recognition_amd/lib/liblas_hip_ablsp$m.so python tools/bench_speller.py 2>&1 | grep speller_fwd; done
