# the frame sort as Onesweep (rocPRIM's merge-sort limit at zero) against its default (block sort + merge passes)
mkdir -p gpurun_out/r06
AB_WORKLOADS="c2:cold:12 c2:steady:8 c4:cold:3" timeout 1500 python profiles/ab_probe.py sage-icp_amd/_probe/libsageicp_prev.so sage-icp_amd/_probe/libsageicp_onesweep.so 2>&1 | tee gpurun_out/r06/onesweep_ab.txt
