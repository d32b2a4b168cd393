/* TEST INFRASTRUCTURE: a C99 consumer of include/dptnav.h -- what a maintainer's C / cgo / FFI binding of the boundary sees.
 * Compiled with `gcc -std=c99 -pedantic -Wall -Werror` (no C++ , no HIP headers, no torch) and linked against libdptnav.so.
 * Without a GPU it checks what can be checked on the host: the ABI version, that dptnav_create refuses bad configurations
 * with a message (and, on a box without a HIP device, a good one too -- the library has no CPU path), and that the size
 * queries are NULL-safe.  With a device (argv[1] = "gpu") it walks the weight table of a one-block model. */
#include <stdio.h>
#include <string.h>

#include "dptnav.h"

static int fail(const char* what) {
  fprintf(stderr, "FAILED: %s\n", what);
  return 1;
}

int main(int argc, char** argv) {
  dptnav_config cfg;
  dptnav_handle h = NULL;
  int rc;
  if (dptnav_abi_version() != DPTNAV_ABI_VERSION) return fail("ABI version of the library differs from the header's");
  memset(&cfg, 0, sizeof cfg);
  cfg.num_features = 96; /* unsupported */
  cfg.video_emb_size = 512; cfg.hidden_video = 96; cfg.kernel_size_enc = 7; cfg.hidden_dim = 128; cfg.num_blocks = 1;
  cfg.chunk_size = 150; cfg.step_size = 75; cfg.num_heads = 4; cfg.bidir = 1;
  rc = dptnav_create(&cfg, &h);
  if (rc != DPTNAV_ERR_INVALID || h != NULL) return fail("num_features = 96 must be refused");
  if (strstr(dptnav_last_error(NULL), "num_features") == NULL) return fail("the refusal must say why");
  if (dptnav_create(NULL, &h) == DPTNAV_OK) return fail("NULL config accepted");
  if (dptnav_num_weights(NULL) != 0 || dptnav_frames(NULL, 32000) != -1 || dptnav_workspace_bytes(NULL, 1, 32000, 50) != 0)
    return fail("size queries must be NULL-safe");
  cfg.num_features = 128; cfg.hidden_video = 128;
  rc = dptnav_create(&cfg, &h);
  if (argc > 1 && strcmp(argv[1], "gpu") == 0) {
    int n, i;
    long long total = 0;
    if (rc != DPTNAV_OK) { fprintf(stderr, "%s\n", dptnav_last_error(NULL)); return fail("dptnav_create on a GPU box"); }
    n = dptnav_num_weights(h);
    if (n != 6 + 2 * 18 + 6) return fail("weight table of a one-block DPTN-AV model: 48 tensors");
    for (i = 0; i < n; ++i) {
      if (dptnav_weight_name(h, i) == NULL || dptnav_weight_numel(h, i) <= 0) return fail("weight table entry");
      total += dptnav_weight_numel(h, i);
    }
    if (strcmp(dptnav_weight_name(h, 0), "gate") != 0) return fail("slot 0 is `gate` (state_dict order)");
    if (dptnav_frames(h, 32000) != 10665 || dptnav_chunks(h, 32000) != 141) return fail("frame / chunk arithmetic");
    if (dptnav_workspace_bytes(h, 16, 32000, 50) == 0) return fail("workspace size");
    if (dptnav_forward(h, NULL, NULL, NULL, 1, 32000, 50, NULL, NULL, NULL, 0, NULL) != DPTNAV_ERR_WEIGHTS)
      return fail("forward before dptnav_bind_weights must fail with DPTNAV_ERR_WEIGHTS");
    printf("ok gpu: %d weight tensors, %lld parameters\n", n, total);
    dptnav_destroy(h);
  } else {
    if (rc == DPTNAV_OK) dptnav_destroy(h);          /* a GPU happens to be visible: fine */
    else if (strstr(dptnav_last_error(NULL), "no CPU path") == NULL) return fail("without a device: say that there is no CPU path");
    printf("ok host\n");
  }
  return 0;
}
