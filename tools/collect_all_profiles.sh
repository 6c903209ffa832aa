set -e
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r3B --config B > gpurun_out/collect_r3B.log 2>&1
echo B done
bash tools/collect_profiles.sh r3C --config C --bytes 4000000000 > gpurun_out/collect_r3C.log 2>&1
echo C done
bash tools/collect_profiles.sh r3D --config D > gpurun_out/collect_r3D.log 2>&1
echo D done
bash tools/collect_profiles.sh r3E --config E --bytes 8000000000 > gpurun_out/collect_r3E.log 2>&1
echo E done
