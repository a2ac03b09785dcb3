# Floor experiments on the folded int8 kernel (wrong-result builds, timing only): which part of a round costs what.
#   bash tools/fold_floor.sh build   (in the build container: variants into build/v_*/, only conv_i8s.hip differs)
#   bash tools/fold_floor.sh run     (on the GPU box: times each variant in its own process)
set -e
VARS="base NOFOLD NOMMA NOEPI NOFOLD_NOEPI NOMMA_NOEPI"
flags() { case $1 in base) echo "" ;; *) echo "$1" | sed 's/^/-DSN_I8F_/; s/_NO/ -DSN_I8F_NO/g' ;; esac; }
if [ "$1" = build ]; then
  make -j8 > /dev/null
  for v in $VARS; do
    mkdir -p build/v_$v build/obj_$v
    for f in cabi bank voxel conv conv_i8 conv_lin backward corr loss; do cp -u build/obj/$f.o build/obj_$v/; done
    touch build/obj_$v/*.o

    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Iscene-net_amd/csrc $(flags $v) -c scene-net_amd/csrc/conv_i8s.hip -o build/obj_$v/conv_i8s.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build/obj_$v/*.o -o build/v_$v/libscenenet_hip.so
    echo built $v "$(flags $v)"
  done
else
  O=gpurun_out/floor; mkdir -p $O
  for v in $VARS; do
    SN_HIP_LIB=$PWD/build/v_$v/libscenenet_hip.so python tools/conv_ab.py --rounds 2 > $O/$v.txt 2>&1
    echo "== $v"; grep -E "^folded" $O/$v.txt
  done
fi
