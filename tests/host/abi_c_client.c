/* A plain-C client of include/steppingstone.h: proves the header is C (not C++) and that the drop-in boundary is usable
 * without Python or torch.  Loads the library with dlopen, checks the version, and asks for an environment; on a box
 * without a GPU ss_create must fail loudly with SS_ERR_NO_DEVICE (there is no CPU fallback). */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "../../include/steppingstone.h"

typedef int (*version_fn)(void);
typedef int (*create_fn)(ss_env**, int, int32_t, int, uint64_t, int64_t);
typedef const char* (*err_fn)(void);
typedef void (*destroy_fn)(ss_env*);

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s libsteppingstone.so\n", argv[0]); return 2; }
  void* h = dlopen(argv[1], RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
  version_fn version = (version_fn)dlsym(h, "ss_version");
  create_fn create = (create_fn)dlsym(h, "ss_create");
  err_fn last_error = (err_fn)dlsym(h, "ss_last_error");
  destroy_fn destroy = (destroy_fn)dlsym(h, "ss_destroy");
  if (!version || !create || !last_error || !destroy) { fprintf(stderr, "missing symbol\n"); return 4; }
  ss_env* env = NULL;
  int rc = create(&env, SS_WALKER3D, 64, 0, 1u, 0);
  printf("version %d create rc %d obs_dim %d act_dim %d msg \"%s\"\n", version(), rc, SS_OBS_DIM, SS_ACT_DIM, rc ? last_error() : "");
  if (rc == SS_OK) destroy(env);
  return 0;
}
