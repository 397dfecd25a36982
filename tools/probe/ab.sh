# ab.sh "<python cmd>" name1 name2 ... : run the command once per variant library (default build = "main")
cmd=$1; shift
for v in "$@"; do
  if [ "$v" = main ]; then unset ZS3_LIB; else export ZS3_LIB=$GRAFT_REPO_ROOT/zs3_amd/lib/variants/libzs3hip_$v.so; fi
  echo "=== $v"
  eval "$cmd" 2>&1 | grep -v amdgpu.ids | tail -${AB_TAIL:-4}
done
