// Matrix-pipe micro-benchmark: a register-only stream of v_mfma_f64_16x16x4_f64 with hand-picked operand
// registers (one asm statement per kernel), to see whether the A/B/C register pattern of the GEMM engine's
// inner product limits the issue rate.  Two wavefronts per SIMD (grid = 512 blocks of 256 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

#define CLOB_V "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15", \
 "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
 "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
 "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63", \
 "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
 "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
 "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111", \
 "v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127", \
 "v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143"
#define CLOB_A "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15", \
 "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31", \
 "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47", \
 "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63", \
 "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79", \
 "a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95", \
 "a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111", \
 "a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127"

// the 16 MFMAs of one k-step; ACC(i) names accumulator i, AOP(i)/BOP(j) the operand pairs
#define STEP(M) M(0,0) M(0,1) M(0,2) M(0,3) M(1,0) M(1,1) M(1,2) M(1,3) M(2,0) M(2,1) M(2,2) M(2,3) M(3,0) M(3,1) M(3,2) M(3,3)

#define S_(x) #x
#define S(x) S_(x)
// K0: every MFMA reads the same A and B pair
#define M_SAME(i,j)  "v_mfma_f64_16x16x4_f64 v[%c[b" S(i) S(j) "]:%c[e" S(i) S(j) "]], v[0:1], v[2:3], v[%c[b" S(i) S(j) "]:%c[e" S(i) S(j) "]]\n\t"

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(int iters, double* out) {
  int cnt = iters;
  if (MODE == 0) {       // same A/B registers, accumulators in VGPRs
    asm volatile(
      "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\t"
      "1:\n\t"
      "v_mfma_f64_16x16x4_f64 v[16:23], v[0:1], v[2:3], v[16:23]\n\t"   "v_mfma_f64_16x16x4_f64 v[24:31], v[0:1], v[2:3], v[24:31]\n\t"
      "v_mfma_f64_16x16x4_f64 v[32:39], v[0:1], v[2:3], v[32:39]\n\t"   "v_mfma_f64_16x16x4_f64 v[40:47], v[0:1], v[2:3], v[40:47]\n\t"
      "v_mfma_f64_16x16x4_f64 v[48:55], v[0:1], v[2:3], v[48:55]\n\t"   "v_mfma_f64_16x16x4_f64 v[56:63], v[0:1], v[2:3], v[56:63]\n\t"
      "v_mfma_f64_16x16x4_f64 v[64:71], v[0:1], v[2:3], v[64:71]\n\t"   "v_mfma_f64_16x16x4_f64 v[72:79], v[0:1], v[2:3], v[72:79]\n\t"
      "v_mfma_f64_16x16x4_f64 v[80:87], v[0:1], v[2:3], v[80:87]\n\t"   "v_mfma_f64_16x16x4_f64 v[88:95], v[0:1], v[2:3], v[88:95]\n\t"
      "v_mfma_f64_16x16x4_f64 v[96:103], v[0:1], v[2:3], v[96:103]\n\t" "v_mfma_f64_16x16x4_f64 v[104:111], v[0:1], v[2:3], v[104:111]\n\t"
      "v_mfma_f64_16x16x4_f64 v[112:119], v[0:1], v[2:3], v[112:119]\n\t" "v_mfma_f64_16x16x4_f64 v[120:127], v[0:1], v[2:3], v[120:127]\n\t"
      "v_mfma_f64_16x16x4_f64 v[128:135], v[0:1], v[2:3], v[128:135]\n\t" "v_mfma_f64_16x16x4_f64 v[136:143], v[0:1], v[2:3], v[136:143]\n\t"
      "s_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1b\n\ts_nop 15\n\ts_nop 15\n\t"
      : "+s"(cnt) : : CLOB_V, "scc");
  } else if (MODE == 1) {  // the engine's pattern: 4 A pairs x 4 B pairs, accumulators in VGPRs
    asm volatile(
      "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
      "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t"
      "1:\n\t"
      "v_mfma_f64_16x16x4_f64 v[16:23], v[0:1], v[8:9], v[16:23]\n\t"   "v_mfma_f64_16x16x4_f64 v[24:31], v[0:1], v[10:11], v[24:31]\n\t"
      "v_mfma_f64_16x16x4_f64 v[32:39], v[0:1], v[12:13], v[32:39]\n\t"   "v_mfma_f64_16x16x4_f64 v[40:47], v[0:1], v[14:15], v[40:47]\n\t"
      "v_mfma_f64_16x16x4_f64 v[48:55], v[2:3], v[8:9], v[48:55]\n\t"   "v_mfma_f64_16x16x4_f64 v[56:63], v[2:3], v[10:11], v[56:63]\n\t"
      "v_mfma_f64_16x16x4_f64 v[64:71], v[2:3], v[12:13], v[64:71]\n\t"   "v_mfma_f64_16x16x4_f64 v[72:79], v[2:3], v[14:15], v[72:79]\n\t"
      "v_mfma_f64_16x16x4_f64 v[80:87], v[4:5], v[8:9], v[80:87]\n\t"   "v_mfma_f64_16x16x4_f64 v[88:95], v[4:5], v[10:11], v[88:95]\n\t"
      "v_mfma_f64_16x16x4_f64 v[96:103], v[4:5], v[12:13], v[96:103]\n\t" "v_mfma_f64_16x16x4_f64 v[104:111], v[4:5], v[14:15], v[104:111]\n\t"
      "v_mfma_f64_16x16x4_f64 v[112:119], v[6:7], v[8:9], v[112:119]\n\t" "v_mfma_f64_16x16x4_f64 v[120:127], v[6:7], v[10:11], v[120:127]\n\t"
      "v_mfma_f64_16x16x4_f64 v[128:135], v[6:7], v[12:13], v[128:135]\n\t" "v_mfma_f64_16x16x4_f64 v[136:143], v[6:7], v[14:15], v[136:143]\n\t"
      "s_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1b\n\ts_nop 15\n\ts_nop 15\n\t"
      : "+s"(cnt) : : CLOB_V, "scc");
  } else if (MODE == 2) {  // the engine's pattern, accumulators in AGPRs
    asm volatile(
      "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
      "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t"
      "1:\n\t"
      "v_mfma_f64_16x16x4_f64 a[0:7], v[0:1], v[8:9], a[0:7]\n\t"   "v_mfma_f64_16x16x4_f64 a[8:15], v[0:1], v[10:11], a[8:15]\n\t"
      "v_mfma_f64_16x16x4_f64 a[16:23], v[0:1], v[12:13], a[16:23]\n\t"   "v_mfma_f64_16x16x4_f64 a[24:31], v[0:1], v[14:15], a[24:31]\n\t"
      "v_mfma_f64_16x16x4_f64 a[32:39], v[2:3], v[8:9], a[32:39]\n\t"   "v_mfma_f64_16x16x4_f64 a[40:47], v[2:3], v[10:11], a[40:47]\n\t"
      "v_mfma_f64_16x16x4_f64 a[48:55], v[2:3], v[12:13], a[48:55]\n\t"   "v_mfma_f64_16x16x4_f64 a[56:63], v[2:3], v[14:15], a[56:63]\n\t"
      "v_mfma_f64_16x16x4_f64 a[64:71], v[4:5], v[8:9], a[64:71]\n\t"   "v_mfma_f64_16x16x4_f64 a[72:79], v[4:5], v[10:11], a[72:79]\n\t"
      "v_mfma_f64_16x16x4_f64 a[80:87], v[4:5], v[12:13], a[80:87]\n\t" "v_mfma_f64_16x16x4_f64 a[88:95], v[4:5], v[14:15], a[88:95]\n\t"
      "v_mfma_f64_16x16x4_f64 a[96:103], v[6:7], v[8:9], a[96:103]\n\t" "v_mfma_f64_16x16x4_f64 a[104:111], v[6:7], v[10:11], a[104:111]\n\t"
      "v_mfma_f64_16x16x4_f64 a[112:119], v[6:7], v[12:13], a[112:119]\n\t" "v_mfma_f64_16x16x4_f64 a[120:127], v[6:7], v[14:15], a[120:127]\n\t"
      "s_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1b\n\ts_nop 15\n\ts_nop 15\n\t"
      : "+s"(cnt) : : CLOB_V, CLOB_A, "scc");
  } else if (MODE == 3) {  // engine pattern in VGPRs + 8 LDS fragment reads per k-step into a second register set
    asm volatile(
      "v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0\n\tv_mov_b32 v5, 0\n\tv_mov_b32 v6, 0\n\tv_mov_b32 v7, 0\n\t"
      "v_mov_b32 v8, 0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\t"
      "v_lshlrev_b32 v144, 3, %1\n\t"
      "1:\n\t"
      "ds_read_b64 v[146:147], v144\n\tds_read_b64 v[148:149], v144 offset:2304\n\tds_read_b64 v[150:151], v144 offset:4608\n\tds_read_b64 v[152:153], v144 offset:6912\n\t"
      "ds_read_b64 v[154:155], v144 offset:18432\n\tds_read_b64 v[156:157], v144 offset:18688\n\tds_read_b64 v[158:159], v144 offset:18944\n\tds_read_b64 v[160:161], v144 offset:19200\n\t"
      "v_mfma_f64_16x16x4_f64 v[16:23], v[0:1], v[8:9], v[16:23]\n\t"   "v_mfma_f64_16x16x4_f64 v[24:31], v[0:1], v[10:11], v[24:31]\n\t"
      "v_mfma_f64_16x16x4_f64 v[32:39], v[0:1], v[12:13], v[32:39]\n\t"   "v_mfma_f64_16x16x4_f64 v[40:47], v[0:1], v[14:15], v[40:47]\n\t"
      "v_mfma_f64_16x16x4_f64 v[48:55], v[2:3], v[8:9], v[48:55]\n\t"   "v_mfma_f64_16x16x4_f64 v[56:63], v[2:3], v[10:11], v[56:63]\n\t"
      "v_mfma_f64_16x16x4_f64 v[64:71], v[2:3], v[12:13], v[64:71]\n\t"   "v_mfma_f64_16x16x4_f64 v[72:79], v[2:3], v[14:15], v[72:79]\n\t"
      "v_mfma_f64_16x16x4_f64 v[80:87], v[4:5], v[8:9], v[80:87]\n\t"   "v_mfma_f64_16x16x4_f64 v[88:95], v[4:5], v[10:11], v[88:95]\n\t"
      "v_mfma_f64_16x16x4_f64 v[96:103], v[4:5], v[12:13], v[96:103]\n\t" "v_mfma_f64_16x16x4_f64 v[104:111], v[4:5], v[14:15], v[104:111]\n\t"
      "v_mfma_f64_16x16x4_f64 v[112:119], v[6:7], v[8:9], v[112:119]\n\t" "v_mfma_f64_16x16x4_f64 v[120:127], v[6:7], v[10:11], v[120:127]\n\t"
      "v_mfma_f64_16x16x4_f64 v[128:135], v[6:7], v[12:13], v[128:135]\n\t" "v_mfma_f64_16x16x4_f64 v[136:143], v[6:7], v[14:15], v[136:143]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_sub_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1b\n\ts_nop 15\n\ts_nop 15\n\t"
      : "+s"(cnt) : "v"((int)threadIdx.x & 63) : CLOB_V, "v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161", "scc", "memory");
  }
  if (cnt == 12345) out[0] = 1.0;
}

template <int MODE>
void run(const char* name) {
  double* d; hipMalloc(&d, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, grid = 512;
  const size_t lds = 40960;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, 2000, d);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, iters, d);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 16 * 2048.0;
  printf("%-64s %.3f ms  %.1f TFLOP/s  (%s)\n", name, ms, flops / ms * 1e-9, hipGetErrorString(hipGetLastError()));
  hipFree(d);
}

int main() {
  run<0>("same A/B pair, C in VGPRs");
  run<1>("4 A x 4 B pairs, C in VGPRs");
  run<2>("4 A x 4 B pairs, C in AGPRs");
  run<3>("4 A x 4 B pairs, C in VGPRs, 8 ds_read_b64 per k-step");
  run<0>("same A/B pair, C in VGPRs (again)");
  return 0;
}
