# FindG2O.cmake -- find-module that satisfies sparse-gslam's `find_package(G2O REQUIRED)`
# (src/sparse_gslam/CMakeLists.txt:34) with the MI355X backend of this repository instead of g2o.
#
# Use: put this directory in front of the reference's own cmake/ on CMAKE_MODULE_PATH (or replace
# src/sparse_gslam/cmake/FindG2O.cmake with this file) and pass -DSGO_ROOT=/path/to/this/repo.
# It defines every variable the reference's CMakeLists consumes (:235-243, :274):
#   G2O_FOUND, G2O_INCLUDE_DIR,
#   G2O_STUFF_LIBRARY, G2O_CORE_LIBRARY, G2O_TYPES_SLAM2D,
#   G2O_SOLVER_SLAM2D_LINEAR, G2O_SOLVER_STRUCTURE_ONLY, G2O_SOLVER_EIGEN
# The g2o-compatible C++ surface is header-only (include/g2o/...); the single library behind it is
# libsgo.so, so G2O_CORE_LIBRARY points at it and the other library variables are left empty
# (an empty entry in target_link_libraries is ignored).

if(NOT SGO_ROOT)
  get_filename_component(SGO_ROOT "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
endif()

find_path(G2O_INCLUDE_DIR g2o/core/base_vertex.h
  PATHS ${SGO_ROOT}/include NO_DEFAULT_PATH)
find_library(G2O_CORE_LIBRARY NAMES sgo
  PATHS ${SGO_ROOT}/sparse_gslam_amd/csrc NO_DEFAULT_PATH)

set(G2O_STUFF_LIBRARY "${G2O_CORE_LIBRARY}")
set(G2O_TYPES_SLAM2D "")
set(G2O_SOLVER_SLAM2D_LINEAR "")
set(G2O_SOLVER_STRUCTURE_ONLY "")
set(G2O_SOLVER_EIGEN "")
set(G2O_SOLVERS_FOUND "YES")

set(G2O_FOUND "NO")
if(G2O_INCLUDE_DIR AND G2O_CORE_LIBRARY)
  set(G2O_FOUND "YES")
  set(G2O_INCLUDE_DIRS ${G2O_INCLUDE_DIR})
  include_directories(BEFORE ${G2O_INCLUDE_DIR})   # g2o/... now resolves to the compat headers
  # libsgo.so needs the HIP runtime at load time
  get_filename_component(_sgo_libdir "${G2O_CORE_LIBRARY}" DIRECTORY)
  list(APPEND CMAKE_BUILD_RPATH "${_sgo_libdir}" "/opt/rocm/lib")
  list(APPEND CMAKE_INSTALL_RPATH "${_sgo_libdir}" "/opt/rocm/lib")
  link_directories(/opt/rocm/lib)
elseif(G2O_FIND_REQUIRED)
  message(FATAL_ERROR "sgo backend not found: build it (make -C ${SGO_ROOT}/sparse_gslam_amd/csrc) or set SGO_ROOT")
endif()
