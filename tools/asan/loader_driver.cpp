#include <cstdio>
#include <cstdint>
#include "../../include/pinot_host_c.h"
int main(int argc, char** argv) {
  for (int i = 1; i < argc; ++i) {
    int32_t st = 0;
    void* seg = ph_segment_load_directory(argv[i], -1, &st);
    if (!seg) { printf("%s: status %d %s\n", argv[i], st, ph_last_error()); continue; }
    char* d = ph_segment_describe(seg, &st);
    printf("%s: ok %.80s\n", argv[i], d ? d : "?");
    if (d) ph_free(d);
    ph_segment_destroy(seg);
  }
  return 0;
}
