# Lab builds of the library with a SEQ_EXP timing switch in k_block_x6<.., SEQ> (results are then WRONG): tools/seqlab_<n>.so.
# usage: bash tools/seq_lab.sh 1 2 3 ...   (run here, in the build container; the .so files travel to the GPU box)
set -e
cd "$(dirname "$0")/.."
OBJ=influentialrs_amd/csrc/_obj
for n in "$@"; do
  if [ "$n" = stamp ]; then DEF="-DX6_STAMP=2"; elif [ "$n" = stamp_builtin_dma ]; then DEF="-DX6_STAMP=2 -DSEQ_ASM_DMA=0"; elif [ "$n" = stamp1 ]; then DEF="-DX6_STAMP=1"; elif [ "$n" = builtin_dma ]; then DEF="-DSEQ_ASM_DMA=0"; else DEF="-DSEQ_EXP=$n"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DIRS_LAB $DEF -c influentialrs_amd/csrc/decoder.hip -o /tmp/decoder_seqlab_$n.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/seqlab_$n.so $OBJ/capi.o /tmp/decoder_seqlab_$n.o $OBJ/score.o $OBJ/path.o $OBJ/comm.o -ldl
  echo built tools/seqlab_$n.so
done
