/* placeholder TU: matcher policies (SearchForInitialization, ComputeStereoMatches) are added here. */
#include "orb_oracle.h"
