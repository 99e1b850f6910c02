// include/shim/boost/optional.hpp — boost::optional / boost::none (src/dqn.hpp:32-33,
// src/dqn_main.cpp:139-140) on std::optional, for boxes without Boost.
#ifndef DQNHIP_SHIM_BOOST_OPTIONAL_HPP_
#define DQNHIP_SHIM_BOOST_OPTIONAL_HPP_
#include <optional>
namespace boost {
template <class T> using optional = std::optional<T>;
inline constexpr std::nullopt_t none{std::nullopt};
}  // namespace boost
#endif
