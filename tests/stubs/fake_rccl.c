/* Test double for librccl (tests/test_comm_stub.py, tests/test_dp_gpu.py): the five entry points vg_comm.hip binds with
 * dlsym, with the call sequence recorded so that a CPU-only test can check the order and arguments of the library's
 * RCCL calls.  With FAKE_RCCL_DIR set the all-reduce is REAL: the ranks (separate processes) exchange their buffers
 * through files in that directory, every rank sums them in rank order (fp32 accumulation, bf16 results rounded to
 * nearest even) and applies the average -- so two ranks can run the hip.comm=abi path end to end and be checked for
 * VALUES, on host buffers (CPU tests) or on device buffers of ranks that share one GPU (real RCCL cannot put two ranks
 * on one device).  Device buffers go through hipMemcpy of the HIP runtime the process already loaded; the call
 * synchronises `stream` first and is blocking, which is a legal (if slow) execution of a stream-ordered collective.
 * Built by the test with gcc; selected through VG_RCCL_LIB.  Not part of the product. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;
typedef int ncclResult_t;

static char g_log[4096];
static char g_last_id[128];
static int g_live = 0;
static int g_rank = 0, g_nranks = 1;
static long g_seq = 0;

static void note(const char* s) {
  if (strlen(g_log) + strlen(s) + 2 < sizeof(g_log)) { strcat(g_log, s); strcat(g_log, ";"); }
}
static int failing(const char* what) {
  const char* e = getenv("FAKE_RCCL_FAIL");
  return e && strcmp(e, what) == 0;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  note("GetUniqueId");
  for (int i = 0; i < 128; ++i) id->internal[i] = (char)(i * 7 + 3);
  return failing("id") ? 2 : 0;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  char b[64];
  strcpy(b, "CommInitRank:");
  b[13] = (char)('0' + rank); b[14] = '/'; b[15] = (char)('0' + nranks); b[16] = 0;
  note(b);
  memcpy(g_last_id, id.internal, 128);
  if (failing("init")) return 3;
  *comm = (ncclComm_t)&g_live;
  g_live = 1;
  g_rank = rank;
  g_nranks = nranks;
  g_seq = 0;
  return 0;
}

/* ---- the file-backed all-reduce (FAKE_RCCL_DIR) */
typedef int (*memcpy_fn)(void*, const void*, size_t, int);
typedef int (*sync_fn)(void*);
static memcpy_fn g_hip_memcpy = NULL;
static sync_fn g_hip_sync = NULL;
static void bind_hip(void) {
  static int tried = 0;
  if (tried) return;
  tried = 1;
  void* h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);      /* only a runtime the process already uses */
  if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("libamdhip64.so.6", RTLD_NOW | RTLD_NOLOAD);
  if (!h) return;
  g_hip_memcpy = (memcpy_fn)dlsym(h, "hipMemcpy");
  g_hip_sync = (sync_fn)dlsym(h, "hipStreamSynchronize");
}
static int copy_any(void* dst, const void* src, size_t bytes) {
  /* device or host pointers alike through the runtime when it works; a box without a GPU (the CPU tests: the runtime is
     loaded but has no device) only ever passes host buffers */
  if (g_hip_memcpy && g_hip_memcpy(dst, src, bytes, 4 /* hipMemcpyDefault */) == 0) return 0;
  memcpy(dst, src, bytes);
  return 0;
}
static float bf16_to_f32(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f32_to_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   /* NaN stays NaN */
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static int wait_for(const char* path) {
  for (int i = 0; i < 600000; ++i) {          /* <= 120 s */
    if (access(path, R_OK) == 0) return 0;
    usleep(200);
  }
  return 1;
}
static int file_allreduce(const char* dir, const void* s, void* r, size_t n, int dtype, int op, void* stream) {
  const size_t esz = dtype == 7 ? 4 : 2, bytes = n * esz;
  char path[1024], tmp[1024];
  bind_hip();
  if (g_hip_sync) (void)g_hip_sync(stream);
  void* mine = malloc(bytes);
  float* acc = (float*)malloc(n * sizeof(float));
  void* other = malloc(bytes);
  if (!mine || !acc || !other) return 2;
  if (copy_any(mine, s, bytes) != 0) return 2;
  snprintf(tmp, sizeof(tmp), "%s/s%ld_r%d.tmp", dir, g_seq, g_rank);
  snprintf(path, sizeof(path), "%s/s%ld_r%d.bin", dir, g_seq, g_rank);
  FILE* f = fopen(tmp, "wb");
  if (!f || fwrite(mine, 1, bytes, f) != bytes) return 2;
  fclose(f);
  if (rename(tmp, path) != 0) return 2;
  for (size_t i = 0; i < n; ++i) acc[i] = 0.f;
  for (int q = 0; q < g_nranks; ++q) {        /* rank order: every rank computes the same bits */
    const void* src = mine;
    if (q != g_rank) {
      snprintf(path, sizeof(path), "%s/s%ld_r%d.bin", dir, g_seq, q);
      if (wait_for(path)) return 2;
      f = fopen(path, "rb");
      if (!f || fread(other, 1, bytes, f) != bytes) return 2;
      fclose(f);
      src = other;
    }
    if (dtype == 7) for (size_t i = 0; i < n; ++i) acc[i] += ((const float*)src)[i];
    else for (size_t i = 0; i < n; ++i) acc[i] += bf16_to_f32(((const uint16_t*)src)[i]);
  }
  const float scale = op == 4 ? 1.0f / (float)g_nranks : 1.0f;
  if (dtype == 7) for (size_t i = 0; i < n; ++i) ((float*)mine)[i] = acc[i] * scale;
  else for (size_t i = 0; i < n; ++i) ((uint16_t*)mine)[i] = f32_to_bf16(acc[i] * scale);
  if (copy_any(r, mine, bytes) != 0) return 2;
  /* everyone has read everyone's buffer once all the .done marks exist: only then may a rank remove its file */
  snprintf(path, sizeof(path), "%s/s%ld_r%d.done", dir, g_seq, g_rank);
  f = fopen(path, "wb");
  if (f) fclose(f);
  for (int q = 0; q < g_nranks; ++q) {
    snprintf(path, sizeof(path), "%s/s%ld_r%d.done", dir, g_seq, q);
    if (wait_for(path)) return 2;
  }
  snprintf(path, sizeof(path), "%s/s%ld_r%d.bin", dir, g_seq, g_rank);
  unlink(path);
  ++g_seq;
  free(mine); free(acc); free(other);
  return 0;
}
ncclResult_t ncclAllReduce(const void* s, void* r, size_t n, int dtype, int op, ncclComm_t c, void* stream) {
  (void)stream;
  char b[96];
  /* dtype: ncclFloat32 = 7, ncclBfloat16 = 9; op: ncclSum = 0, ncclAvg = 4 */
  strcpy(b, "AllReduce:");
  b[10] = (char)('0' + dtype); b[11] = ','; b[12] = (char)('0' + op); b[13] = ',';
  b[14] = (s == r) ? 'I' : 'O'; b[15] = ','; b[16] = (c == (ncclComm_t)&g_live && g_live) ? 'L' : 'D'; b[17] = 0;
  note(b);
  if (failing("allreduce")) return 5;
  const char* dir = getenv("FAKE_RCCL_DIR");
  if (dir && g_nranks > 1 && (dtype == 7 || dtype == 9) && (op == 0 || op == 4))
    return file_allreduce(dir, s, r, n, dtype, op, stream);
  return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  note(c == (ncclComm_t)&g_live ? "CommDestroy" : "CommDestroy:BAD");
  g_live = 0;
  return failing("destroy") ? 1 : 0;
}
const char* ncclGetErrorString(ncclResult_t rc) { return rc == 5 ? "fake rccl: all-reduce refused" : "fake rccl: error"; }

/* test hooks */
const char* fake_rccl_log(void) { return g_log; }
const char* fake_rccl_last_id(void) { return g_last_id; }
