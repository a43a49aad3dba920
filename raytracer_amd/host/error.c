/* Text for the errno codes of the host layer (src/error.h:19-42). */
#include "rt_host.h"

#include <string.h>

char *rtGetErrorDesc(void) {
  static char text[256];
  const char *m;
  switch (errno) {
    case E_IO: m = "unable to read or write from/to file"; break;
    case E_MEMORY: m = "unable to allocate memory"; break;
    case E_INVALID_FILE: m = "trying to load file of invalid format"; break;
    case E_INVALID_FILE_FORMAT: m = "image file format is not supported"; break;
    case E_INVALID_BPP: m = "bits per pixel argument is invalid (correct ones are: 24, 32)"; break;
    case E_NOT_ENOUGH_SURFACES: m = "not enough surfaces to cover entire scene"; break;
    case E_INVALID_PARAM_VALUE: m = "parameter has invalid value"; break;
    default: m = strerror(errno); break;
  }
  strncpy(text, m, sizeof text - 1);
  text[sizeof text - 1] = 0;
  return text;
}
