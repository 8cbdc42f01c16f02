cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3s
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3s/trace -- python3 tools/post_profile.py > gpurun_out/r3s/post.json 2> gpurun_out/r3s/post.err; echo "rc=$?"
cat gpurun_out/r3s/post.json; grep -h "frbch" gpurun_out/r3s/trace/*/*kernel_stats.csv | cut -c1-200
