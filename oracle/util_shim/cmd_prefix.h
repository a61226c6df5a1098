// TEST INFRASTRUCTURE ONLY (oracle): force-included before the reference's cmd/main.cpp (which relies on its util
// headers pulling these in).
#include <cmath>
#include <iostream>
#include <string>
