// placeholder until the prefix-beam kernel lands (symbols must exist for the ABI check)
#include "common.h"
extern "C" size_t ms_ctc_beam_workspace_bytes(int, int, int, int) { return 0; }
extern "C" int ms_ctc_beam_decode(const float*, const int32_t*, int32_t*, int32_t*, int, int, int, int, int, float, int,
                                  const float*, int, int, const float*, int, int32_t*, int32_t*, int32_t*, void*, size_t,
                                  void*) {
  ms::set_error("ms_ctc_beam_decode: not built yet");
  return MS_ERR_UNSUPPORTED;
}
