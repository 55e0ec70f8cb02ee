for c in "lanczos 1280 720 1600 900" "lanczos 1280 720 1920 1080" "lanczos 1280 720 1920 1080 RGB" "lanczos 1920 1080 1278 718"; do
  tag=$(echo $c | tr ' ' '_')
  bash tools/prof_pmc.sh $tag "python /root/repo/tools/resize_one.py $c" > gpurun_out/prof_$tag.txt 2>&1
  echo "== $c"; grep -E "^void|SQ_INSTS_(VALU|SALU|LDS|BRANCH|SMEM|VMEM_RD)|ACTIVE_INST_(ANY|VALU|SCA|LDS|MISC|VMEM)|SQ_WAVES " gpurun_out/prof_$tag.txt | awk '{print $1, $NF}'; tail -3 gpurun_out/prof_$tag.txt | head -2 | cut -c1-120
done
