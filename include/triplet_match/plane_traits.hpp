/* drop-in name of the reference header include/plane_traits */
#include "traits.hpp"
