#define IGEMM_WS_IS1X1 true
#define IGEMM_WS_ENTRY conv_igemm_ws_1x1
#include "igemm_ws.inc"
