// fy_itemsim.hip -- item-item similarity build (placeholder until the row kernel's top-K epilogue lands).
#include "fy_rm2.hpp"
namespace fy {
fy_result* itemsim_build(Context*, const fy_itemsim_params*, const fy_ratings*) {
    FY_FAIL(FY_ERR_UNSUPPORTED, "item-item similarity build is not implemented yet");
}
}  // namespace fy
