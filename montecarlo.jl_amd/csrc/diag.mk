# Diagnostic builds that are not part of the product (kept out of the Makefile so that they do not touch the kernel
# source hash): make -C montecarlo.jl_amd/csrc -f diag.mk lu4_stamps [STAMP_LEVEL=2]
include Makefile
lu4_stamps: $(OBJS)
	$(HIPCC) $(FLAGS) -DLU4_STAMPS=$(STAMP_LEVEL) -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -pragma-unroll-threshold=4000000 --cuda-device-only -S sweep_lu.hip -o sweep_lu_stamps.dev.s
	$(WAR) --patch sweep_lu_stamps.dev.s sweep_lu_stamps.guard.s
	$(LLVM)/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=$(ARCH) -c sweep_lu_stamps.guard.s -o sweep_lu_stamps.dev.o
	$(LLVM)/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o sweep_lu_stamps.hsaco sweep_lu_stamps.dev.o
	$(LLVM)/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$(ARCH) -input=/dev/null -input=sweep_lu_stamps.hsaco -output=sweep_lu_stamps.hipfb
	$(HIPCC) $(FLAGS) -DLU4_STAMPS=$(STAMP_LEVEL) -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -pragma-unroll-threshold=4000000 --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang sweep_lu_stamps.hipfb -c sweep_lu.hip -o sweep_lu_stamps.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o ../libdqmc_hip_lu4stamps.so $(filter-out sweep_lu.o,$(OBJS)) sweep_lu_stamps.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
# the one-launch UDT with stamps (tools/qrb_stamps.py); the stamps build does not compile at every revision
qrb_stamps: $(OBJS)
	$(HIPCC) $(FLAGS) -DQRB_STAMPS -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -pragma-unroll-threshold=4000000 --cuda-device-only -S qrb.hip -o qrb_stamps.dev.s
	$(WAR) --patch qrb_stamps.dev.s qrb_stamps.guard.s
	$(LLVM)/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=$(ARCH) -c qrb_stamps.guard.s -o qrb_stamps.dev.o
	$(LLVM)/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o qrb_stamps.hsaco qrb_stamps.dev.o
	$(LLVM)/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$(ARCH) -input=/dev/null -input=qrb_stamps.hsaco -output=qrb_stamps.hipfb
	$(HIPCC) $(FLAGS) -DQRB_STAMPS -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -pragma-unroll-threshold=4000000 --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang qrb_stamps.hipfb -c qrb.hip -o qrb_stamps.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o ../libdqmc_hip_qrbstamps.so $(filter-out qrb.o,$(OBJS)) qrb_stamps.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
