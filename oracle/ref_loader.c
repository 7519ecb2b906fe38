/*
 * ref_loader.c — opens oracle/_ref/*.so with RTLD_LAZY.
 *
 * TEST INFRASTRUCTURE ONLY.  Python's ctypes always adds RTLD_NOW, which
 * would try to bind the reference's unresolved rtlsdr_* function references
 * (librtlsdr/libusb are not in this image and no stand-ins are written for
 * them).  Lazy binding leaves them untouched because the DSP path never calls
 * them.  The returned handle is wrapped with ctypes.CDLL(None, handle=...).
 */
#include <dlfcn.h>
#include <stdio.h>

void *ref_loader_open(const char *path)
{
	void *h = dlopen(path, RTLD_LAZY | RTLD_LOCAL);
	if (!h)
		fprintf(stderr, "ref_loader_open(%s): %s\n", path, dlerror());
	return h;
}

int ref_loader_close(void *handle)
{
	return handle ? dlclose(handle) : -1;
}

void *ref_loader_sym(void *handle, const char *name)
{
	return dlsym(handle, name);
}
