// hgs_io.h — socket helpers shared by hnsw_gpu_server (server_main.cpp) and its client library
// (remote_client.cpp): framed messages of include/hnsw_gpu_server.h over a Unix stream socket,
// optional file descriptor (memfd) passed with SCM_RIGHTS.
#pragma once
#include <cerrno>
#include <cstdint>
#include <cstring>
#include <poll.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <sys/uio.h>
#include <unistd.h>

#include "hnsw_gpu_server.h"

namespace hgs {

// Send header + up to two payload pieces (+ one fd with the first byte).  Works on blocking and
// non-blocking sockets (waits for POLLOUT, `timeout_ms` per wait).  0 on success, -1 on error.
inline int send_msg(int fd, const hgs_hdr *h, const void *p1, size_t l1, const void *p2, size_t l2,
					int pass_fd = -1, int timeout_ms = 10000)
{
	struct iovec iov[3];
	iov[0].iov_base = const_cast<hgs_hdr *>(h); iov[0].iov_len = sizeof(*h);
	iov[1].iov_base = const_cast<void *>(p1);   iov[1].iov_len = p1 ? l1 : 0;
	iov[2].iov_base = const_cast<void *>(p2);   iov[2].iov_len = p2 ? l2 : 0;
	int first = 0;
	bool fd_sent = pass_fd < 0;
	while (first < 3)
	{
		if (iov[first].iov_len == 0) { first++; continue; }
		struct msghdr mh;
		memset(&mh, 0, sizeof(mh));
		mh.msg_iov = iov + first;
		mh.msg_iovlen = (size_t) (3 - first);
		alignas(struct cmsghdr) char cbuf[CMSG_SPACE(sizeof(int))];
		if (!fd_sent)
		{
			memset(cbuf, 0, sizeof(cbuf));
			mh.msg_control = cbuf;
			mh.msg_controllen = sizeof(cbuf);
			struct cmsghdr *cm = CMSG_FIRSTHDR(&mh);
			cm->cmsg_level = SOL_SOCKET;
			cm->cmsg_type = SCM_RIGHTS;
			cm->cmsg_len = CMSG_LEN(sizeof(int));
			memcpy(CMSG_DATA(cm), &pass_fd, sizeof(int));
		}
		ssize_t n = sendmsg(fd, &mh, MSG_NOSIGNAL);
		if (n < 0)
		{
			if (errno == EINTR) continue;
			if (errno == EAGAIN || errno == EWOULDBLOCK)
			{
				struct pollfd pf = { fd, POLLOUT, 0 };
				int pr = poll(&pf, 1, timeout_ms);
				if (pr > 0) continue;
				if (pr < 0 && errno == EINTR) continue;
				return -1;
			}
			return -1;
		}
		fd_sent = true;
		size_t left = (size_t) n;
		while (left > 0 && first < 3)
		{
			if (left >= iov[first].iov_len) { left -= iov[first].iov_len; iov[first].iov_len = 0; first++; }
			else { iov[first].iov_base = (char *) iov[first].iov_base + left; iov[first].iov_len -= left; left = 0; }
		}
	}
	return 0;
}

// Blocking read of exactly n bytes; a descriptor that arrives on the way is stored in *got_fd
// (extra ones are closed).  0 on success, -1 on error / EOF / nothing for `timeout_ms` (< 0 = wait for ever):
// a backend must not hang for good on a server that has stopped answering.
inline int recv_exact(int fd, void *buf, size_t n, int *got_fd, int timeout_ms = -1)
{
	char *p = (char *) buf;
	while (n > 0)
	{
		if (timeout_ms >= 0)
		{
			struct pollfd pf = { fd, POLLIN, 0 };
			const int pr = poll(&pf, 1, timeout_ms);
			if (pr == 0) { errno = ETIMEDOUT; return -1; }
			if (pr < 0) { if (errno == EINTR) continue; return -1; }
		}
		struct iovec iov = { p, n };
		struct msghdr mh;
		memset(&mh, 0, sizeof(mh));
		mh.msg_iov = &iov;
		mh.msg_iovlen = 1;
		alignas(struct cmsghdr) char cbuf[CMSG_SPACE(4 * sizeof(int))];
		mh.msg_control = cbuf;
		mh.msg_controllen = sizeof(cbuf);
		ssize_t r = recvmsg(fd, &mh, MSG_CMSG_CLOEXEC);
		if (r < 0)
		{
			if (errno == EINTR) continue;
			return -1;
		}
		if (r == 0) return -1;
		for (struct cmsghdr *cm = CMSG_FIRSTHDR(&mh); cm; cm = CMSG_NXTHDR(&mh, cm))
			if (cm->cmsg_level == SOL_SOCKET && cm->cmsg_type == SCM_RIGHTS)
			{
				size_t cnt = (cm->cmsg_len - CMSG_LEN(0)) / sizeof(int);
				for (size_t i = 0; i < cnt; i++)
				{
					int f;
					memcpy(&f, CMSG_DATA(cm) + i * sizeof(int), sizeof(int));
					if (got_fd && *got_fd < 0) *got_fd = f; else close(f);
				}
			}
		p += r;
		n -= (size_t) r;
	}
	return 0;
}

}  // namespace hgs
