/* drop-in name of the reference header include/cylinder_traits */
#include "traits.hpp"
