#include "caffe2/core/context.h"   // oracle shim: one header holds the whole stand-in
