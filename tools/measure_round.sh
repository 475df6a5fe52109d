#!/bin/bash
# Full measurement pass of a round (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats, PMC passes.
# usage: bash tools/measure_round.sh <subdir of gpurun_out>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-final}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --extra-file $O/bench_extra.json > $O/bench_default.json 2> $O/bench_default.err   # the driver's line (compact) + the long form: extra.{batch_sweep, arithmetic_*, car_90_320_3, ...}
for b in 1 64 128 512; do
  python3 $R/bench.py --batch $b --no-cpu-baseline --no-extras > $O/bench_b$b.json 2> /dev/null
done
python3 $R/bench.py --batch 512 --arithmetic pieces --no-cpu-baseline --no-extras > $O/bench_pieces.json 2> /dev/null
python3 $R/bench.py --batch 512 --arithmetic f32 --no-cpu-baseline --no-extras --steps 20 > $O/bench_f32.json 2> /dev/null
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 $R/bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_torchrun.json 2> $O/bench_torchrun.err
python3 $R/tools/l1_split_check.py --max-obj 500 --batch 64 128 256 512 --steps 10 --arithmetic f16x2 > $O/l1_check_f16x2.json 2> /dev/null
python3 $R/tools/l1_split_check.py --max-obj 500 --batch 64 128 256 512 --steps 10 --arithmetic f32 > $O/l1_check_f32.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o d -- python3 $R/bench.py --batch 512 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1024 -o d -- python3 $R/bench.py --batch 1024 --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_b1024.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b1 -o d -- python3 $R/bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_b1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b64 -o d -- python3 $R/bench.py --batch 64 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_b64.log 2>&1
# K0 (round 4): accuracy + timing against MIOpen, kernel stats and counters of the fp16 kernel at 8 frame pairs (one head, seven heads)
(cd $R && python3 tools/conv_check.py --batches 1,8 > $O/conv_check.jsonl 2> /dev/null)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_conv_b8 -o d -- python3 $R/tools/conv_only.py --batch 8 --iters 10 > $O/prof_conv_b8.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_conv_b8_heads7 -o d -- python3 $R/tools/conv_only.py --batch 8 --iters 6 --heads 7 > $O/prof_conv_b8_heads7.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_convmfma_b8 -o p -- python3 $R/tools/conv_only.py --batch 8 --iters 4 > $O/pmc_convmfma_b8.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_convlds_b8 -o p -- python3 $R/tools/conv_only.py --batch 8 --iters 4 > $O/pmc_convlds_b8.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_convfetch_b8 -o p -- python3 $R/tools/conv_only.py --batch 8 --iters 4 > $O/pmc_convfetch_b8.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_convwrite_b8 -o p -- python3 $R/tools/conv_only.py --batch 8 --iters 4 > $O/pmc_convwrite_b8.log 2>&1
# pair stage at F = 320 (round 4, pair_f16w.hip): kernel stats at the car tables x 512 and at N = 500 x 256, matrix / vector pipe counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pair320_car -o d -- python3 $R/tools/pair_time.py --max-obj 90 --batch 512 --iters 20 > $O/pair320_car.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pair320_n500 -o d -- python3 $R/tools/pair_time.py --max-obj 500 --batch 256 --iters 10 > $O/pair320_n500.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_pair320mfma_b256 -o p -- python3 $R/tools/pair_time.py --max-obj 500 --batch 256 --iters 3 --modes f16x2 > $O/pmc_pair320mfma.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_pair320valu_b256 -o p -- python3 $R/tools/pair_time.py --max-obj 500 --batch 256 --iters 3 --modes f16x2 > $O/pmc_pair320valu.log 2>&1
# the configs 2-4 chain (round 4)
(cd $R && timeout 300 python3 tools/time_pipeline.py --scenes 20 --frames 40 --batch 40 --sync 1 > $O/pipeline_sync.log 2>&1; timeout 300 python3 tools/time_pipeline.py --scenes 20 --frames 40 --batch 40 --sync 0 > $O/pipeline.log 2>&1)
for b in 1 128 512 1024; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_b$b -o p -- python3 $R/bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_fetch_b$b.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_b$b -o p -- python3 $R/bench.py --batch $b --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_write_b$b.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_b1024 -o p -- python3 $R/bench.py --batch 1024 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_mfma_b1024.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_b512 -o p -- python3 $R/bench.py --batch 512 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_mfma_b512.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/pmc_hit_b512 -o p -- python3 $R/bench.py --batch 512 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_hit_b512.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_valu_b512 -o p -- python3 $R/bench.py --batch 512 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/pmc_valu_b512.log 2>&1
(cd $R && python3 tools/stage_power.py 512 2 > $O/stage_power.log 2>&1)
# round 5: the one-pass aff kernels (phase stamps of a workgroup, both arithmetics), K0 forms (conv_check prints 1 head / 7 heads)
if [ -x $R/tools/probes/_bin/affq128 ]; then
  (AFF_F16=1 $R/tools/probes/_bin/affq128 1024 500 2; $R/tools/probes/_bin/affq128 1024 500 2) > $O/aff_frame_probe.log 2>&1
fi
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_convmfma7_b8 -o p -- python3 $R/tools/conv_only.py --batch 8 --iters 4 --heads 7 > $O/pmc_convmfma7_b8.log 2>&1
# round 5: the training step (kernel stats at N=500 B=8 and the car configuration x 64; the per-pair MLP kernels alone)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_n500 -o d -- python3 $R/tools/time_train.py --max-obj 500 --feats 7 --points 4 --batch 8 --steps 8 --in-backward > $O/train_n500.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train_n90 -o d -- python3 $R/tools/time_train.py --max-obj 90 --feats 3 --points 5 --batch 64 --steps 8 --in-backward > $O/train_n90.log 2>&1
(cd $R && python3 tools/time_pair_mlp.py > $O/pair_mlp.log 2>&1; python3 tools/time_adam_lowrank.py >> $O/pair_mlp.log 2>&1; python3 tools/time_adam_lowrank.py --dx >> $O/pair_mlp.log 2>&1; python3 tools/time_adam_lowrank.py --rank 64 >> $O/pair_mlp.log 2>&1)
# round 6: K0 in train() mode (kernel stats + counters of the weight-gradient kernel at 8 frame pairs), K1 one cloud / a batch of 16
bash $R/tools/gpu_pmc_convtrain.sh ${1:-final} 8 > $O/convtrain_pmc.log 2>&1
(cd $R && python3 tools/time_conv_train.py --batch 8 > $O/conv_train.jsonl 2>/dev/null; python3 tools/time_conv_train.py --batch 2 >> $O/conv_train.jsonl 2>/dev/null; python3 tools/time_conv_train.py --batch 64 --iters 4 >> $O/conv_train.jsonl 2>/dev/null)
VOX_ITERS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_voxelize -o d -- python3 $R/tools/time_voxelize.py > $O/voxelize_prof.log 2>&1
(cd $R && python3 tools/time_voxelize.py > $O/voxelize.log 2>&1)
(cd $R && python3 tools/time_gather_bwd.py 2>&1 | grep us > $O/gather_bwd.log; python3 tools/probes/train_determinism.py 2>&1 | tail -3 > $O/train_determinism.log; python3 tools/train_soak_conv.py 2>&1 | tail -3 > $O/train_soak_conv.log)
grep -h '^{' $O/bench_default.json $O/bench_b1.json $O/bench_b64.json $O/bench_b128.json $O/bench_b512.json $O/bench_pieces.json $O/bench_f32.json $O/bench_torchrun.json | cut -c1-230
grep time $O/conv_check.jsonl | cut -c1-200; tail -1 $O/pipeline.log | cut -c1-400; grep -h "^{" $O/pair320_car.log $O/pair320_n500.log
ls $O
# K1 HBM traffic (round 6): FETCH_SIZE / WRITE_SIZE of the voxeliser's kernels, one cloud and the batch of 16
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_voxfetch_b16 -o p -- python3 $R/tools/time_voxelize.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_voxwrite_b16 -o p -- python3 $R/tools/time_voxelize.py > /dev/null 2>&1
