#!/bin/bash
# Collect the per-round measurement evidence on the GPU box (run through gpurun from the repo root):
#   tools/collect_evidence.sh r5        -> gpurun_out/<prefix>_{bench.json,phases.txt,kernel_stats.csv,pmc.csv,pmc.json,phase_stamps.txt,phase_stamps_insitu.txt,
#                                          timeline.txt,bucket_sweep.txt,decode_{bench.json,kernel_stats.csv,pmc.csv,pmc.json},speller_phase_stamps.txt,
#                                          speller_loc_phase_stamps.txt,bench_config3.json,config3_kernel_stats.csv}
# (build first: make -C automatic-speech-recognition_amd/csrc all prof; hipcc ... -DLAS_ROW_STAMPS tools/micro/bench_fused.hip -o tools/micro/bin/bench_fused_stamps)
# rocprofv3 databases go to /tmp (they exceed gpurun's 64 MiB merge limit); only the summaries are kept.  Counter passes are
# separate runs with --kernel-trace only (MI355X_MICROARCH.md, HBM section); python3 bench.py directly after `--`.
set -u
# (build both libraries first: make -C automatic-speech-recognition_amd/csrc all prof -- the phase stamps need liblas_hip_prof.so of the SAME source)
P=${1:-r2}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
# ---- first, on the fresh box: the whole GPU suite as the driver runs it (one process, -rs: every skip with its reason).  (Round 5: run LAST, behind eleven
# profiler passes and the eight-process host-time tool, it hit the documented exchange time-out once -- profiles/r5_pytest_gpu_behind_profilers.log)
LAS_PARITY_LOG=$PWD/gpurun_out/${P}_parity_full_T.jsonl python3 -m pytest tests -m gpu -q -rs > gpurun_out/${P}_pytest_gpu.log 2>&1; tail -4 gpurun_out/${P}_pytest_gpu.log
rocprofv3 --kernel-trace --stats -d /tmp/kt_$P -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-train-loop > gpurun_out/${P}_kt.log 2>&1
python3 tools/kernel_stats.py /tmp/kt_$P 3 gpurun_out/${P}_kernel_stats.csv > /dev/null
# counter passes serialise kernels: the x-projection chunks (another stream's kernels the running sweep waits for) must be off there
export LAS_ALLOW_SERIAL_STREAMS=1      # (r4: the SAME kernel instances as the timed step -- chunk products, chunk-aware sweeps, CH = true BPTT -- with every producer in front of its consumer)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${P}_$c -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode --no-train-loop > /tmp/pmc_${P}_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  -d /tmp/pmc_${P}_SQ -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode --no-train-loop > /tmp/pmc_${P}_SQ.log 2>&1
python3 tools/pmc_summary.py gpurun_out/$P /tmp/pmc_${P}_FETCH_SIZE /tmp/pmc_${P}_WRITE_SIZE /tmp/pmc_${P}_SQ > gpurun_out/${P}_pmc_summary.log 2>&1
unset LAS_ALLOW_SERIAL_STREAMS
# the bench line comes AFTER the counter passes: bench.py takes the dominant kernel's HBM traffic from the newest profiles/*_pmc.json
# that was recorded from this very csrc/rnn_seq.hip
cp gpurun_out/${P}_pmc.json profiles/${P}_pmc.json 2>/dev/null
LAS_PHASES=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --no-train-loop > /dev/null 2> gpurun_out/${P}_phases.txt   # spans of the phases / sweeps (HIP events; their recording costs ~0.1 ms per step)
python3 tools/prof_rnn.py > gpurun_out/${P}_phase_stamps.txt 2>&1
python3 tools/prof_rnn_insitu.py > gpurun_out/${P}_phase_stamps_insitu.txt 2>&1          # the last BPTT sweep's split inside a whole step
python3 tools/timeline.py "$(ls /tmp/kt_$P/*/*_results.db /tmp/kt_$P/*_results.db 2>/dev/null | head -1)" --list > gpurun_out/${P}_timeline.txt 2>&1
# ---- decode leg (BASELINE configs[4]: beam 16 + 2x512 char RNNLM, 16 utterances x 16 beams = 256 rows per step): its own kernel
# trace, counter passes (HBM bytes, MFMA busy, waits) and bench object with the per-part timing / roofline
rocprofv3 --kernel-trace --stats -d /tmp/kt_dec_$P -o b -- python3 bench.py --decode-only > gpurun_out/${P}_decode_kt.log 2>&1
python3 tools/rocpd_summary.py "$(ls /tmp/kt_dec_$P/*/*_results.db /tmp/kt_dec_$P/*_results.db 2>/dev/null | head -1)" gpurun_out/${P}_decode_kernel_stats.csv > /dev/null 2>&1
export LAS_ALLOW_SERIAL_STREAMS=1      # (the encoders' chunked x-projections: see above)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_dec_${P}_$c -o p -- python3 bench.py --decode-only > /tmp/pmc_dec_${P}_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  -d /tmp/pmc_dec_${P}_SQ -o p -- python3 bench.py --decode-only > /tmp/pmc_dec_${P}_SQ.log 2>&1
python3 tools/pmc_summary.py gpurun_out/${P}_decode /tmp/pmc_dec_${P}_FETCH_SIZE /tmp/pmc_dec_${P}_WRITE_SIZE /tmp/pmc_dec_${P}_SQ > gpurun_out/${P}_decode_pmc_summary.log 2>&1
unset LAS_ALLOW_SERIAL_STREAMS
cp gpurun_out/${P}_decode_pmc.json profiles/${P}_decode_pmc.json 2>/dev/null          # (decode.roofline.traffic: the newest profiles/*_decode_pmc.json recorded from this csrc/speller.hip)
python3 bench.py --decode-only > gpurun_out/${P}_decode_bench.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 > gpurun_out/${P}_bench.json 2> /dev/null             # the bench line exactly as the driver runs it (behind BOTH counter passes it quotes)
# ---- Speller loop kernels: phase stamps of one decode step (row workgroup 0 and product workgroup 0 on one clock)
[ -x tools/micro/bin/bench_fused_stamps ] && tools/micro/bin/bench_fused_stamps > gpurun_out/${P}_speller_phase_stamps.txt 2>&1
[ -x tools/micro/bin/bench_fused_stamps ] && tools/micro/bin/bench_fused_stamps 0 loc > gpurun_out/${P}_speller_loc_phase_stamps.txt 2>&1
# ---- the second configuration (BASELINE configs[3]: V = 5000 + location-aware attention) on one rank
rocprofv3 --kernel-trace --stats -d /tmp/kt_c3_$P -o b -- python3 bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-train-loop > gpurun_out/${P}_c3_kt.log 2>&1
python3 tools/kernel_stats.py /tmp/kt_c3_$P 3 gpurun_out/${P}_config3_kernel_stats.csv > /dev/null
python3 bench.py --config 3 --no-decode --no-train-loop > gpurun_out/${P}_bench_config3.json 2> /dev/null
python3 tools/bucket_sweep.py > gpurun_out/${P}_bucket_sweep.txt 2>&1
# ---- round 4: the parity mode (exact-fp32 MFMA kernels) -- its own kernel trace and the GEMM shapes against the 157 TF/s fp32 peak
rocprofv3 --kernel-trace --stats -d /tmp/kt_f32_$P -o b -- python3 bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-decode --no-train-loop > gpurun_out/${P}_f32_kt.log 2>&1
python3 tools/kernel_stats.py /tmp/kt_f32_$P 1 gpurun_out/${P}_f32_kernel_stats.csv > /dev/null
python3 tools/bench_gemm_f32.py > gpurun_out/${P}_f32_gemm_shapes.txt 2>&1
python3 bench.py --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-train-loop > gpurun_out/${P}_bench_f32.json 2> /dev/null
PROF_PREC=0 python3 tools/prof_rnn.py > gpurun_out/${P}_f32_phase_stamps.txt 2>&1        # round 5: the exact-fp32 cluster sweeps' phases
[ -x tools/micro/bin/bench_mfma_f32 ] && tools/micro/bin/bench_mfma_f32 > gpurun_out/${P}_mfma_f32_rate.txt 2>&1
# ---- the cell the reference builds (BasicRNNCell), speed mode: kernel trace
rocprofv3 --kernel-trace --stats -d /tmp/kt_rnn_$P -o b -- python3 bench.py --cell rnn --steps 10 --warmup 3 --no-cpu-baseline --no-decode --no-train-loop > gpurun_out/${P}_rnn_kt.log 2>&1
python3 tools/kernel_stats.py /tmp/kt_rnn_$P 3 gpurun_out/${P}_rnn_kernel_stats.csv > /dev/null
# ---- round 5: the decode step at decode.py's batch (64 utterances x beam 16 = 1024 rows): consecutive kernels of replayed steps
(cd /tmp && NUTT=64 rocprofv3 --kernel-trace --stats -d /tmp/kt_dec64_$P -o b -- python3 $OLDPWD/tools/probe_decode_step.py > $OLDPWD/gpurun_out/${P}_decode_b64_kt.log 2>&1)
python3 tools/rocpd_summary.py "$(find /tmp/kt_dec64_$P -name '*_results.db' | head -1)" gpurun_out/${P}_decode_b64_kernel_stats.csv > /dev/null 2>&1
# ---- round 5: the beam search's LSTM cell launches -- us per launch, the 128-row body's phase stamps (make ablf F=loss_opt D=-DLB_STAMP=1), the per-CU ingest
# rate they are measured against, and a stream of batches with the next batch's encoders under the current search
MS="256 1024" python3 tools/bench_cell_rows.py 2>&1 | grep -v amdgpu > gpurun_out/${P}_cell_rows.txt
[ -f automatic-speech-recognition_amd/lib/liblas_hip_ablf.so ] && MS=1024 LAS_LIB_PATH=automatic-speech-recognition_amd/lib/liblas_hip_ablf.so STAMP=1 python3 tools/bench_cell_rows.py 2>&1 | grep -v amdgpu > gpurun_out/${P}_cell_phase_stamps.txt
[ -x tools/micro/bin/bench_ingest2 ] && tools/micro/bin/bench_ingest2 > gpurun_out/${P}_cu_ingest.txt 2>&1
[ -f automatic-speech-recognition_amd/lib/liblas_hip_beamst.so ] && LAS_LIB_PATH=automatic-speech-recognition_amd/lib/liblas_hip_beamst.so python3 tools/probe_beam_stamps.py 2>&1 | grep -v amdgpu > gpurun_out/${P}_beam_phase_stamps.txt   # make ablf F=beam D=-DLAS_BEAM_STAMPS S=beamst
if [ -f automatic-speech-recognition_amd/lib/liblas_hip_rowst.so ]; then    # make ablf F=speller D=-DLAS_ROW_STAMPS S=rowst
  (LAS_LIB_PATH=automatic-speech-recognition_amd/lib/liblas_hip_rowst.so python3 tools/probe_pf_stamps.py; LAS_LIB_PATH=automatic-speech-recognition_amd/lib/liblas_hip_rowst.so python3 tools/probe_rows4_stamps.py) 2>&1 | grep -v amdgpu > gpurun_out/${P}_rows_phase_stamps.txt
fi
python3 tools/probe_decode_stream.py 2>&1 | grep -v amdgpu > gpurun_out/${P}_decode_stream.txt
# ---- round 5: are small launches on the chain what the profiler says they are?  (prepared sweeps vs self-packing, un-profiled, alternating)
bash tools/ab_bench.sh LAS_NO_PREPARED_SWEEPS=0 LAS_NO_PREPARED_SWEEPS=1 3 60 > gpurun_out/${P}_ab_prepared.txt 2>&1
python3 tools/probe_host_ahead.py > gpurun_out/${P}_host_ahead.txt 2>&1
# ---- eight ranks' host side on this box (one GPU: the steps run one rank at a time behind command-processor gates)
timeout 900 python3 tools/host_time_ranks.py --ranks 8 --steps 4 --out gpurun_out/${P}_host_ranks_8.json > /dev/null 2>&1
# ---- round 6: the reference's own recipe (run.sh:59-76) through the wide Speller path: kernel traces of both cells; the scheduled-sampling leg
for c in rnn lstm; do
  rocprofv3 --kernel-trace --stats -d /tmp/kt_rs_${c}_$P -o b -- python3 bench.py --only-leg run_sh_$c --steps 5 --warmup 2 > gpurun_out/${P}_runsh_${c}_kt.log 2>&1
  python3 tools/kernel_stats.py /tmp/kt_rs_${c}_$P 2 gpurun_out/${P}_runsh_${c}_kernel_stats.csv > /dev/null
done
cp gpurun_out/${P}_runsh_rnn_kernel_stats.csv gpurun_out/${P}_runsh_kernel_stats.csv 2>/dev/null
LAS_NO_WIDE=1 python3 bench.py --only-leg run_sh --steps 3 --warmup 1 > gpurun_out/${P}_runsh_round5_kernels.json 2> /dev/null     # the same leg on round 5's per-utterance rows
python3 bench.py --only-leg run_sh --steps 5 --warmup 2 > gpurun_out/${P}_runsh.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d /tmp/kt_ss_$P -o b -- python3 bench.py --only-leg config2_sampling --steps 10 --warmup 3 > gpurun_out/${P}_sampling_kt.log 2>&1
python3 tools/kernel_stats.py /tmp/kt_ss_$P 3 gpurun_out/${P}_sampling_kernel_stats.csv > /dev/null
python3 bench.py --only-leg config2_sampling --steps 20 --warmup 5 > gpurun_out/${P}_sampling.json 2> /dev/null
# ---- round 6: run-to-run determinism over the schedule knobs / token sources at the geometries where the duplicate stores were found, and the
# sweeps repeated on identical inputs (alone and beside memory traffic on another stream)
(for b in 5 8 16; do echo "B=$b"; B=$b python3 tools/probe_determinism.py; done; echo "bench geometry"; BIG=1 FEW=1 python3 tools/probe_determinism.py) 2>&1 | grep -v amdgpu > gpurun_out/${P}_determinism.txt
(for b in 5 8 24; do CONC=1 B=$b python3 tools/probe_sweep_repeat.py; done) 2>&1 | grep -v amdgpu > gpurun_out/${P}_sweep_repeat.txt
# ---- round 6: an utterance's hypothesis rows on one XCD (las_speller_fwd_args.row_group) against row = workgroup id, alternating runs
(for i in 1 2 3; do for v in 0 1; do echo -n "LAS_NO_XCD_LOCAL_ROWS=$v "; LAS_NO_XCD_LOCAL_ROWS=$v python3 bench.py --decode-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())['decode']
print('value', d['value'], 'value_b16', d['value_b16'], 'us_per_step', d['us_per_decode_step'], d['step_parts_us'])"; done; done) > gpurun_out/${P}_decode_xcd_ab.txt 2>&1
# ---- round 6: ONE keys / encoder block per utterance for its hypothesis rows (LAS_SPELLER_SHARED_OPERANDS) against a tiled copy per row, alternating
(for i in 1 2 3; do for v in 1 0; do echo -n "LAS_NO_SHARED_OPERANDS=$v "; LAS_NO_SHARED_OPERANDS=$v python3 bench.py --decode-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())['decode']
print('value', d['value'], 'value_b16', d['value_b16'], 'us_per_step', d['us_per_decode_step'], d['step_parts_us'])"; done; done) > gpurun_out/${P}_decode_shared_ab.txt 2>&1
# ---- round 6: the wide Speller path's attention kernels: phase stamps (make ablf F=speller D=-DLAS_ROW_STAMPS S=rowst), the fused launches against one
# launch per phase, the listener's tanh sweeps on two / four members per direction (run.sh recipe, alternating)
if [ -f automatic-speech-recognition_amd/lib/liblas_hip_rowst.so ]; then
  (for c in rnn lstm; do CELL=$c LAS_LIB_PATH=automatic-speech-recognition_amd/lib/liblas_hip_rowst.so python3 tools/probe_wide_stamps.py; done) 2>&1 | grep -v amdgpu > gpurun_out/${P}_wide_phase_stamps.txt
fi
(for i in 1 2 3; do for v in 1 0; do echo -n "LAS_NO_FUSED_STEP=$v "; LAS_NO_FUSED_STEP=$v python3 bench.py --only-leg run_sh 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())['run_sh']
print({k: v['ms_per_step'] for k, v in r.items()})"; done; done) > gpurun_out/${P}_runsh_fused_ab.txt 2>&1
(for i in 1 2 3; do for v in 2 4; do echo -n "LAS_SEQ_P=$v "; LAS_SEQ_P=$v python3 bench.py --only-leg run_sh_rnn 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())['run_sh_rnn']
print({k: v['ms_per_step'] for k, v in r.items()})"; done; done) > gpurun_out/${P}_runsh_cluster_width_ab.txt 2>&1
# ---- round 6: what the step recovery's bookkeeping costs the headline (alternating)
(for i in 1 2 3; do for v in 0 1; do echo -n "LAS_NO_STEP_RECOVERY=$v "; LAS_NO_STEP_RECOVERY=$v python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-decode --no-train-loop --no-side-legs 2>/dev/null | python3 -c "
import json,sys
print(json.loads(sys.stdin.read())['ms_per_step'])"; done; done) > gpurun_out/${P}_ab_recovery.txt 2>&1
tail -c 600 gpurun_out/${P}_bench.json; echo; tail -3 gpurun_out/${P}_pmc_summary.log | cut -c1-300
