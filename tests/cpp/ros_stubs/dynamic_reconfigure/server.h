#pragma once
#include <functional>
namespace dynamic_reconfigure {
template <typename C> struct Server { typedef std::function<void(C&, uint32_t)> CallbackType; void setCallback(const CallbackType&); };
}
