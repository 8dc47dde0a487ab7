#define IGEMM_WS_IS1X1 false
#define IGEMM_WS_ENTRY conv_igemm_ws_gen
#include "igemm_ws.inc"
