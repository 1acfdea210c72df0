// he355_internal.h — in-library access to the parameter object behind a he355_ctx (not part of the C ABI)
#pragma once
#include "../../include/he355.h"
#include "he_params.h"
const he355::Params *he355_internal_params(const he355_ctx *ctx);
