/* drop-in name of the reference header include/identity_traits */
#include "traits.hpp"
