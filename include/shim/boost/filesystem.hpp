// include/shim/boost/filesystem.hpp — boost::filesystem (path, is_regular_file, remove:
// src/dqn_main.cpp:7, 15, 225-243, 403; src/dqn.cpp:95-96) on std::filesystem, for boxes
// without Boost.
#ifndef DQNHIP_SHIM_BOOST_FILESYSTEM_HPP_
#define DQNHIP_SHIM_BOOST_FILESYSTEM_HPP_
#include <filesystem>
namespace boost { namespace filesystem = std::filesystem; }
#endif
