/* Test double for librccl (tests/test_comm_stub.py): the five entry points vg_comm.hip binds with dlsym, with the
 * call sequence recorded so that a CPU-only test can check the order and arguments of the library's RCCL calls.
 * Built by the test with gcc; selected through VG_RCCL_LIB.  Not part of the product. */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef void* ncclComm_t;
typedef int ncclResult_t;

static char g_log[4096];
static char g_last_id[128];
static int g_live = 0;

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
  (void)n;
  return failing("allreduce") ? 5 : 0;
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
