# asm of ONE kernel instantiation: bash scratch/dev_one.sh "false,false,false,0" out.s [extra -D flags]
cd /root/repo/transformer_engine_amd/csrc
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I../../include -I. -fno-gpu-rdc -DTE_GEMM_PART=0 "-DTE_GEMM_DEV_ONE=$1" $3 $4 --cuda-device-only -S gemm_fp8.hip -o $2 2>&1 | grep -v "hip-link"
grep -c "vmcnt(0) lgkmcnt(6)" $2
