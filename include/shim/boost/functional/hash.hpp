// include/shim/boost/functional/hash.hpp — included by src/dqn.hpp:11; nothing of it is used.
#ifndef DQNHIP_SHIM_BOOST_FUNCTIONAL_HASH_HPP_
#define DQNHIP_SHIM_BOOST_FUNCTIONAL_HASH_HPP_
#include <functional>
#endif
