// Stage-loop micro-benchmark of the fp64 GEMM engine's inner loop, written as one asm statement so that every
// ingredient can be switched on its own: MFMAs (always), LDS fragment reads (R), global loads (G), LDS refill
// stores (W), barrier (B).  Two workgroups of 256 threads per CU, as the engine runs.  Timing only.
#include <hip/hip_runtime.h>
#include <cstdio>
#define MFMA16 "v_mfma_f64_16x16x4_f64 v[16:23], v[0:1], v[8:9], v[16:23]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[24:31], v[0:1], v[10:11], v[24:31]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[32:39], v[0:1], v[12:13], v[32:39]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[40:47], v[0:1], v[14:15], v[40:47]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[48:55], v[2:3], v[8:9], v[48:55]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[56:63], v[2:3], v[10:11], v[56:63]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[64:71], v[2:3], v[12:13], v[64:71]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[72:79], v[2:3], v[14:15], v[72:79]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[80:87], v[4:5], v[8:9], v[80:87]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[88:95], v[4:5], v[10:11], v[88:95]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[96:103], v[4:5], v[12:13], v[96:103]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[104:111], v[4:5], v[14:15], v[104:111]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[112:119], v[6:7], v[8:9], v[112:119]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[120:127], v[6:7], v[10:11], v[120:127]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[128:135], v[6:7], v[12:13], v[128:135]\n\t" \
      "v_mfma_f64_16x16x4_f64 v[136:143], v[6:7], v[14:15], v[136:143]\n\t"
#define READS8 "ds_read_b64 v[146:147], v144\n\tds_read_b64 v[148:149], v144 offset:2304\n\tds_read_b64 v[150:151], v144 offset:4608\n\tds_read_b64 v[152:153], v144 offset:6912\n\t" \
      "ds_read_b64 v[154:155], v144 offset:18432\n\tds_read_b64 v[156:157], v144 offset:18688\n\tds_read_b64 v[158:159], v144 offset:18944\n\tds_read_b64 v[160:161], v144 offset:19200\n\t"
#define GLOADS "global_load_dwordx4 v[162:165], v[194:195], off offset:0\n\t" \
      "global_load_dwordx4 v[166:169], v[194:195], off offset:256\n\t" \
      "global_load_dwordx4 v[170:173], v[194:195], off offset:512\n\t" \
      "global_load_dwordx4 v[174:177], v[194:195], off offset:768\n\t" \
      "global_load_dwordx4 v[178:181], v[194:195], off offset:1024\n\t" \
      "global_load_dwordx4 v[182:185], v[194:195], off offset:1280\n\t" \
      "global_load_dwordx4 v[186:189], v[194:195], off offset:1536\n\t" \
      "global_load_dwordx4 v[190:193], v[194:195], off offset:1792\n\t"
#define LWRITES "ds_write_b128 v145, v[162:165] offset:24576\n\t" \
      "ds_write_b128 v145, v[166:169] offset:29184\n\t" \
      "ds_write_b128 v145, v[170:173] offset:33792\n\t" \
      "ds_write_b128 v145, v[174:177] offset:38400\n\t" \
      "ds_write_b128 v145, v[178:181] offset:43008\n\t" \
      "ds_write_b128 v145, v[182:185] offset:47616\n\t" \
      "ds_write_b128 v145, v[186:189] offset:52224\n\t" \
      "ds_write_b128 v145, v[190:193] offset:56832\n\t"
#define GLDS "s_mov_b32 m0, 24576\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[194:195], off offset:0\n\t" \
      "s_mov_b32 m0, 25600\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[194:195], off offset:256\n\t" \
      "s_mov_b32 m0, 26624\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[194:195], off offset:512\n\t" \
      "s_mov_b32 m0, 27648\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[194:195], off offset:768\n\t" \
      "s_mov_b32 m0, 28672\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[194:195], off offset:1024\n\t" \
      "s_mov_b32 m0, 29696\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[194:195], off offset:1280\n\t" \
      "s_mov_b32 m0, 30720\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[194:195], off offset:1536\n\t" \
      "s_mov_b32 m0, 31744\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v[194:195], off offset:1792\n\t"
#define WAITL "s_waitcnt lgkmcnt(0)\n\t"

template <int R, int G, int W, int B, int PIPE, int X = 0>
__global__ __launch_bounds__(256, 2) void k(int iters, const double* src, double* out) {
  extern __shared__ double smem[];
  int cnt = iters;
  const double* gp = src + (blockIdx.x % 64) * 4096 + threadIdx.x * 2;
  asm volatile(
      "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
      "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t"
      "v_lshlrev_b32 v144, 3, %1\n\tv_lshlrev_b32 v145, 4, %2\n\tv_mov_b32 v194, %3\n\tv_mov_b32 v195, %4\n\t" \
      "1:\n\t" \
      ".if %c5 == 1\n\t" GLOADS ".endif\n\t"
      ".if %c5 == 2\n\t" GLDS ".endif\n\t"
      // k-step 0
      ".if %c6\n\t" READS8 ".endif\n\t"
      ".if %c10 == 1\n\t" "v_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\tv_add_u32 v192, v192, 1\n\tv_add_u32 v193, v193, 1\n\tv_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\tv_add_u32 v192, v192, 1\n\tv_add_u32 v193, v193, 1\n\t" ".endif\n\t"
      ".if %c10 == 2\n\t" "v_mul_f64 v[190:191], v[190:191], v[190:191]\n\tv_mul_f64 v[192:193], v[192:193], v[192:193]\n\tv_lshl_add_u64 v[186:187], v[186:187], 0, v[188:189]\n\tv_lshl_add_u64 v[188:189], v[188:189], 0, v[186:187]\n\t" ".endif\n\t"
      ".if %c10 == 3\n\t" "s_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1\n\ts_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1\n\t" ".endif\n\t"
      ".if %c10 == 4\n\t" ".rept 16\n\t" "v_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\t" ".endr\n\t" ".endif\n\t"
      MFMA16
      ".if %c6 && !%c9\n\t" WAITL ".endif\n\t"
      // k-step 1
      ".if %c6\n\t" READS8 ".endif\n\t"
      ".if %c10 == 1\n\t" "v_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\tv_add_u32 v192, v192, 1\n\tv_add_u32 v193, v193, 1\n\tv_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\tv_add_u32 v192, v192, 1\n\tv_add_u32 v193, v193, 1\n\t" ".endif\n\t"
      ".if %c10 == 2\n\t" "v_mul_f64 v[190:191], v[190:191], v[190:191]\n\tv_mul_f64 v[192:193], v[192:193], v[192:193]\n\tv_lshl_add_u64 v[186:187], v[186:187], 0, v[188:189]\n\tv_lshl_add_u64 v[188:189], v[188:189], 0, v[186:187]\n\t" ".endif\n\t"
      ".if %c10 == 3\n\t" "s_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1\n\ts_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1\n\t" ".endif\n\t"
      ".if %c10 == 4\n\t" ".rept 16\n\t" "v_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\t" ".endr\n\t" ".endif\n\t"
      MFMA16
      ".if %c6 && !%c9\n\t" WAITL ".endif\n\t"
      // k-step 2 (refill stores go out before its MFMAs when PIPE, after them otherwise)
      ".if %c6\n\t" READS8 ".endif\n\t"
      ".if %c7 && %c9\n\t" "s_waitcnt vmcnt(0)\n\t" LWRITES ".endif\n\t"
      ".if %c10 == 1\n\t" "v_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\tv_add_u32 v192, v192, 1\n\tv_add_u32 v193, v193, 1\n\tv_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\tv_add_u32 v192, v192, 1\n\tv_add_u32 v193, v193, 1\n\t" ".endif\n\t"
      ".if %c10 == 2\n\t" "v_mul_f64 v[190:191], v[190:191], v[190:191]\n\tv_mul_f64 v[192:193], v[192:193], v[192:193]\n\tv_lshl_add_u64 v[186:187], v[186:187], 0, v[188:189]\n\tv_lshl_add_u64 v[188:189], v[188:189], 0, v[186:187]\n\t" ".endif\n\t"
      ".if %c10 == 3\n\t" "s_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1\n\ts_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1\n\t" ".endif\n\t"
      ".if %c10 == 4\n\t" ".rept 16\n\t" "v_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\t" ".endr\n\t" ".endif\n\t"
      MFMA16
      ".if %c6 && !%c9\n\t" WAITL ".endif\n\t"
      // k-step 3
      ".if %c9\n\t"
        ".if %c5 == 2\n\t" "s_waitcnt vmcnt(0)\n\t" ".endif\n\t"
        ".if %c8\n\t" WAITL "s_barrier\n\t" ".endif\n\t"
        ".if %c6\n\t" READS8 ".endif\n\t"
        ".if %c10 == 1\n\t" "v_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\tv_add_u32 v192, v192, 1\n\tv_add_u32 v193, v193, 1\n\tv_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\tv_add_u32 v192, v192, 1\n\tv_add_u32 v193, v193, 1\n\t" ".endif\n\t"
      ".if %c10 == 2\n\t" "v_mul_f64 v[190:191], v[190:191], v[190:191]\n\tv_mul_f64 v[192:193], v[192:193], v[192:193]\n\tv_lshl_add_u64 v[186:187], v[186:187], 0, v[188:189]\n\tv_lshl_add_u64 v[188:189], v[188:189], 0, v[186:187]\n\t" ".endif\n\t"
      ".if %c10 == 3\n\t" "s_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1\n\ts_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1\n\t" ".endif\n\t"
      ".if %c10 == 4\n\t" ".rept 16\n\t" "v_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\t" ".endr\n\t" ".endif\n\t"
      MFMA16
      ".else\n\t"
        ".if %c6\n\t" READS8 ".endif\n\t"
        ".if %c10 == 1\n\t" "v_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\tv_add_u32 v192, v192, 1\n\tv_add_u32 v193, v193, 1\n\tv_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\tv_add_u32 v192, v192, 1\n\tv_add_u32 v193, v193, 1\n\t" ".endif\n\t"
      ".if %c10 == 2\n\t" "v_mul_f64 v[190:191], v[190:191], v[190:191]\n\tv_mul_f64 v[192:193], v[192:193], v[192:193]\n\tv_lshl_add_u64 v[186:187], v[186:187], 0, v[188:189]\n\tv_lshl_add_u64 v[188:189], v[188:189], 0, v[186:187]\n\t" ".endif\n\t"
      ".if %c10 == 3\n\t" "s_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1\n\ts_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1\n\t" ".endif\n\t"
      ".if %c10 == 4\n\t" ".rept 16\n\t" "v_add_u32 v190, v190, 1\n\tv_add_u32 v191, v191, 1\n\t" ".endr\n\t" ".endif\n\t"
      MFMA16
        ".if %c7\n\t" "s_waitcnt vmcnt(0)\n\t" LWRITES ".endif\n\t"
        ".if %c5 == 2\n\t" "s_waitcnt vmcnt(0)\n\t" ".endif\n\t"
        ".if %c8\n\t" WAITL "s_barrier\n\t" ".endif\n\t"
      ".endif\n\t"
      "s_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1b\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\t"
      : "+s"(cnt)
      : "v"((int)threadIdx.x & 63), "v"((int)threadIdx.x), "v"((unsigned)(uintptr_t)gp), "v"((unsigned)((uintptr_t)gp >> 32)),
        "i"(G), "i"(R), "i"(W), "i"(B), "i"(PIPE), "i"(X)
      : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195", "s40", "s41", "s42", "s43", "scc", "memory");
  if (cnt == 12345) out[0] = smem[0];
}

template <int R, int G, int W, int B, int PIPE, int X = 0>
void run(const char* name, const double* src, double* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 5000, grid = 512;   // one iteration = one k-stage of 64 MFMAs per wavefront
  const size_t lds = 73728;
  hipFuncSetAttribute((const void*)&k<R, G, W, B, PIPE, X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k<R, G, W, B, PIPE, X>), dim3(grid), dim3(256), lds, 0, 500, src, d);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<R, G, W, B, PIPE, X>), dim3(grid), dim3(256), lds, 0, iters, src, d);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 64 * 2048.0;
  printf("%-58s %.3f ms  %.1f TFLOP/s  (%s)\n", name, ms, flops / ms * 1e-9, hipGetErrorString(hipGetLastError()));
}

int main() {
  double *src, *d; hipMalloc(&src, 64 * 4096 * 8 + 65536); hipMalloc(&d, 8);
  hipMemset(src, 0, 64 * 4096 * 8 + 65536);
  run<0, 0, 0, 0, 0>("MFMA only", src, d);
  run<1, 0, 0, 0, 0>("+ fragment reads (wait each k-step)", src, d);
  run<1, 0, 0, 0, 1>("+ fragment reads (no per-step wait)", src, d);
  run<1, 0, 0, 1, 0>("+ reads + barrier", src, d);
  run<1, 1, 0, 0, 0>("+ reads + global loads", src, d);
  run<1, 1, 1, 0, 0>("+ reads + global loads + LDS refill (end of stage)", src, d);
  run<1, 1, 1, 1, 0>("+ reads + loads + refill + barrier  (engine, round 1)", src, d);
  run<1, 1, 1, 1, 1>("+ reads + loads + early refill + barrier before last step", src, d);
  run<0, 1, 1, 1, 0>("MFMA + loads + refill + barrier (no fragment reads)", src, d);
  run<0, 0, 0, 1, 0>("MFMA + barrier", src, d);
  run<1, 2, 0, 1, 0>("+ reads + LDS-DMA loads + barrier (end of stage)", src, d);
  run<1, 2, 0, 1, 1>("+ reads + LDS-DMA loads + barrier before last step", src, d);
  run<1, 2, 0, 0, 0>("+ reads + LDS-DMA loads, no barrier", src, d);
  run<1, 0, 0, 0, 0, 1>("reads + 8 v_add_u32 per k-step", src, d);
  run<1, 0, 0, 0, 0, 4>("reads + 32 v_add_u32 per k-step", src, d);
  run<1, 0, 0, 0, 0, 2>("reads + 2 v_mul_f64 + 2 v_lshl_add_u64 per k-step", src, d);
  run<1, 0, 0, 0, 0, 3>("reads + 8 s_add_u32 per k-step", src, d);
  run<1, 2, 0, 1, 0, 1>("reads + LDS-DMA + barrier + 8 v_add_u32 per k-step", src, d);
  return 0;
}
