R=$PWD/gemini-seal_amd/lib
for r in 1 2 3; do
  for l in libsealhip.so libsealhip_fz1.so; do echo -n "$l "; SEALHIP_LIBRARY=$R/$l python tools/ntt_only.py --logn 15 --polys 8192 | cut -c1-90; done
done
tools/ab_step.sh 1024 libsealhip.so libsealhip_fz1.so | cut -c1-140
