/* test stand-in, see refmodel.h */
#include "refmodel.h"
