/* drop-in name of the reference header include/plane2_traits */
#include "traits.hpp"
